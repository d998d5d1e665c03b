#!/bin/bash
# On the GPU box: DINO64 schedule knobs, each alone (--serial) and pipelined.  usage: tools/sweep_dino.sh [out dir]
out=${1:-gpurun_out/sweep_dino}; mkdir -p $out
run() { # label, args...
  l=$1; shift
  python bench.py --config DINO64 --steps 40 --warmup 4 --no-cpu-baseline --no-check "$@" > $out/$l.json 2>$out/err.txt || tail -3 $out/err.txt
  python - "$l" $out/$l.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2]))
sm=j["config"]["stage_ms"]
print("%-28s ms/step %.3f  stages %s"%(sys.argv[1], j["ms_per_step"], {k[:8]:round(v,3) for k,v in sm.items()}))
PY
}
for rep in 1 2; do
run default
run serial --serial
run cap4M --isect-cap 4000000
run cap4M_serial --isect-cap 4000000 --serial
run side1 --side-streams 1
run side3 --side-streams 3
run prio_off --front-prio off
run prio_on --front-prio on
done 2>&1 | tee $out/table.txt
