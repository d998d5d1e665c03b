#!/bin/bash
# On the GPU box: build the library at other optimisation levels INTO tools/lib/ (make VARIANT=..: the in-tree product
# libgwbp.so is never replaced -- ADVICE r5) and run the differential check and the GPU parity tests against each build.
# The hand-written kernels keep asm-issued memory operations in flight across compiler-visible code; a different register
# allocation is the cheapest way to find a place where that is not watertight (round 5 found one in k_scatter_wide this way).
# Every variant's scatter_wide object passes tools/check_asm_hazards.py --wide inside the Makefile, or the build fails.
# usage: tools/alt_build_test.sh [-O2 -O1 ...]
for opt in "${@:--O2 -O1}"; do
  name=$(echo "$opt" | tr -d '-' | tr 'A-Z' 'a-z')
  make -C 3dgs-gradient-backprojection_amd/csrc -s -j16 VARIANT=$name OPT="$opt" 2>/dev/null || { echo "build failed: $opt"; continue; }
  lib=tools/lib/libgwbp_$name.so
  echo "== $opt wide vs narrow: $(python tools/wide_vs_narrow.py $lib 2>&1 | tail -1)"
  echo "== $opt GPU suite:      $(GWBP_TEST_LIB=$lib python -m pytest tests -m gpu -q -x 2>&1 | tail -1)"
done
