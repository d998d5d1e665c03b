#!/bin/bash
# On the GPU box: one bench config under several pipeline depths / stream counts.  usage: tools/sweep_depth.sh <config> "<depths>" [out dir] [extra args]
C=${1:-DINO64}; out=${3:-gpurun_out/sweep_$C}; mkdir -p $out
for d in ${2:-2 3 4}; do
  for rep in 1 2; do
    python bench.py --config $C --steps 40 --warmup 4 --no-cpu-baseline --no-check --depth $d $4 > $out/d${d}_$rep.json 2>$out/err.txt || tail -3 $out/err.txt
    python3 -c "
import json,sys
j=json.load(open('$out/d${d}_$rep.json')); sm=j['config']['stage_ms']
print('$C depth $d  ms/step %.3f  stages %s' % (j['ms_per_step'], {k[:8]: round(v,3) for k,v in sm.items()}))"
  done
done 2>&1 | tee $out/table.txt
