#!/bin/bash
# Ablations of the 256-channel scatter kernel on the PROFILE library (make -C <pkg>/csrc PROFILE=1):
# GWBP_ABLATE bit0 = plain stores instead of the flush atomics, bit1 = no FMAs, bit2 = no slab staging.  Results invalid by design.
out=${1:-gpurun_out/ablate}; mkdir -p $out
LIB="--lib $PWD/tools/lib/libgwbp_profile.so"
for sched in --serial ""; do
  for ab in 0 1 2 4 3 7; do
    GWBP_ABLATE=$ab python bench.py --steps 30 --no-cpu-baseline --no-check $LIB $sched > $out/ab${ab}${sched}.json 2>/dev/null
    python - <<PY
import json
j=json.load(open("$out/ab${ab}${sched}.json"))
print("ablate=$ab sched='${sched}' ms/step %.3f scatter %.3f front %.3f"%(j["ms_per_step"], j["config"]["stage_ms"]["scatter"], list(j["config"]["stage_ms"].values())[0]))
PY
  done
done
