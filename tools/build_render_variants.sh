#!/bin/bash
# Builds of the library that differ only in render_wide.hip's compile-time knobs.
# usage: tools/build_render_variants.sh name "flags" [name "flags" ...]  ->  tools/lib/libgwbp_<name>.so   (tools/time_render.py D reps config <lib>)
#   e.g. tools/build_render_variants.sh q2s3 "-DGWBP_RENDER_Q2 -DGWBP_RENDER_SLOTS2=3"   (512 channels per wave, three visits in flight)
set -e
cd "$(dirname "$0")/../3dgs-gradient-backprojection_amd/csrc"
make -s -j8
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics"
OTHERS=$(ls *.o | grep -v "prof.o" | grep -v render_wide)
mkdir -p ../../tools/lib
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $FLAGS $2 -c render_wide.hip -o /tmp/render_wide.$1.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lib/libgwbp_$1.so $OTHERS /tmp/render_wide.$1.o
  echo built tools/lib/libgwbp_$1.so
  shift 2
done
