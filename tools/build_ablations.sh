#!/bin/bash
# Structure-preserving ablation builds of k_scatter_wide (compile-time: the counted waits stay intact).
# usage: tools/build_ablations.sh "1 2 3 4 9 16"   ->  tools/lib/libgwbp_abl<bits>.so   (load with bench.py --lib <path>, i.e. _lib.use_library(path, allow_profile=True))
set -e
cd "$(dirname "$0")/../3dgs-gradient-backprojection_amd/csrc"
make -s -j8 PROFILE=1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -DGWBP_PROFILE"
OTHERS=$(ls *.prof.o | grep -v scatter_wide)
for a in ${1:-0 1 2 3 4 9 16}; do
  /opt/rocm/bin/hipcc $FLAGS -DGWBP_ABL=$a -c scatter_wide.hip -o /tmp/scatter_wide.abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lib/libgwbp_abl$a.so $OTHERS /tmp/scatter_wide.abl$a.o
  echo built tools/lib/libgwbp_abl$a.so
done
