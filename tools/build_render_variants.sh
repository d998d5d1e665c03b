#!/bin/bash
# Builds of the library that differ only in compile-time knobs (render_wide.hip's, or any other -D).
# usage: tools/build_render_variants.sh name "flags" [name "flags" ...]  ->  tools/lib/libgwbp_<name>.so   (tools/time_render.py D reps config <lib>)
#   e.g. tools/build_render_variants.sh q2s3 "-DGWBP_RENDER_Q2 -DGWBP_RENDER_SLOTS2=3"   (512 channels per wave, three visits in flight)
set -e
cd "$(dirname "$0")/../3dgs-gradient-backprojection_amd/csrc"
while [ $# -ge 2 ]; do
  make -s -j8 VARIANT=$1 EXTRA="$2"
  echo built tools/lib/libgwbp_$1.so
  shift 2
done
