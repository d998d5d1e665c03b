#!/bin/bash
# On the GPU box: k_scatter_wide ablation table (tools/build_ablations.sh builds), each build alone (--serial) and pipelined.
out=${1:-gpurun_out/r4_abl}; mkdir -p $out
for rep in 1 2; do
for a in ${ABLS:-0 1 9 2 3 4 16}; do
  LIB="--lib $PWD/tools/lib/libgwbp_abl$a.so"
  for sched in --serial ""; do
    python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-check --scatter wide $LIB $sched > $out/abl${a}${sched}_$rep.json 2>$out/err.txt || tail -3 $out/err.txt
    python - "$a" "$sched" $out/abl${a}${sched}_$rep.json <<'PY'
import json,sys
j=json.load(open(sys.argv[3]))
sm=j["config"]["stage_ms"]
print("abl=%-3s %-9s ms/step %.3f  stages %s"%(sys.argv[1], sys.argv[2] or "pipelined", j["ms_per_step"], {k:round(v,3) for k,v in sm.items()}))
PY
  done
done
done 2>&1 | tee $out/table.txt
