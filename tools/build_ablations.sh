#!/bin/bash
# Structure-preserving ablation builds of k_scatter_wide (compile-time: the counted waits stay intact).
# usage: tools/build_ablations.sh "1 2 3 4 9 16"   ->  tools/lib/libgwbp_abl<bits>.so   (load with bench.py --lib <path>, i.e. _lib.use_library(path, allow_profile=True))
# Every variant goes through the Makefile, i.e. through the scatter_wide assembly gate (tools/check_asm_hazards.py --wide).
set -e
cd "$(dirname "$0")/../3dgs-gradient-backprojection_amd/csrc"
for a in ${1:-0 1 2 3 4 9 16}; do
  make -s -j8 VARIANT=abl$a EXTRA="-DGWBP_PROFILE -DGWBP_ABL=$a"
  echo built tools/lib/libgwbp_abl$a.so
done
