#!/bin/bash
# On the GPU box: A/B of library builds under tools/lib/ (name "product" = the in-tree libgwbp.so), each alone (--serial) and pipelined.
# usage: tools/ab_libs.sh "product w6 b64" [out dir] [extra bench args]
out=${2:-gpurun_out/ab}; mkdir -p $out
for rep in 1 2; do
for a in $1; do
  if [ $a = product ]; then LIB=""; else LIB="--lib $PWD/tools/lib/libgwbp_$a.so"; fi
  for sched in --serial ""; do
    python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-check $LIB $3 $sched > $out/${a}${sched}_$rep.json 2>$out/err.txt || tail -3 $out/err.txt
    python - "$a" "$sched" $out/${a}${sched}_$rep.json <<'PY'
import json,sys
j=json.load(open(sys.argv[3]))
sm=j["config"]["stage_ms"]
print("%-10s %-9s ms/step %.3f  stages %s"%(sys.argv[1], sys.argv[2] or "pipelined", j["ms_per_step"], {k[:8]:round(v,3) for k,v in sm.items()}))
PY
  done
done
done 2>&1 | tee $out/table.txt
