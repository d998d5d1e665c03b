#!/bin/bash
# On the GPU box: rebuild the IN-TREE library with other optimisation levels and run the GPU parity tests against each build.
# The hand-written kernels keep asm-issued memory operations in flight across compiler-visible code; a different register
# allocation is the cheapest way to find a place where that is not watertight (round 5 found one in k_scatter_wide this way).
# usage: tools/alt_build_test.sh ["-O2" "-O1" ...]      (the scratch copy of the repository on the box is rebuilt, nothing travels back)
BASE="-std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics"
for opt in "${@:--O2 -O1}"; do
  make -C 3dgs-gradient-backprojection_amd/csrc -s clean
  make -C 3dgs-gradient-backprojection_amd/csrc -s -j16 FLAGS="$opt $BASE" 2>/dev/null || { echo "build failed: $opt"; continue; }
  echo "== $opt: $(python -m pytest tests -m gpu -q -x --deselect tests/test_capi_cpu.py 2>&1 | tail -1)"
done
