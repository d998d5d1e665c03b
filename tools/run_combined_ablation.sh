#!/bin/bash
# On the GPU box (round 5): the COMBINED structure-preserving ablation of k_scatter_wide -- fewer flush
# atomics (bits 1|8 = 9) AND one v_readlane pair per batch (bit 32) together, which round 4 only ever ran apart.
# C2 alone (--serial) and pipelined, then C4 pipelined through the wide kernel.  Builds: tools/build_ablations.sh "0 9 32 41 43 1 33".
out=${1:-gpurun_out/r5_abl}; mkdir -p $out
line() { python - "$1" "$2" "$3" <<'PY'
import json,sys
j=json.load(open(sys.argv[3]))
sm=j["config"]["stage_ms"]
print("abl=%-3s %-9s ms/step %.3f  stages %s"%(sys.argv[1], sys.argv[2] or "pipelined", j["ms_per_step"], {k:round(v,3) for k,v in sm.items()}))
PY
}
{
echo "## C2"
for rep in 1 2; do
for a in ${ABLS:-0 9 32 41 43 1 33}; do
  LIB="--lib $PWD/tools/lib/libgwbp_abl$a.so"
  for sched in --serial ""; do
    python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-check --scatter wide $LIB $sched > $out/c2_abl${a}${sched}_$rep.json 2>$out/err.txt || tail -3 $out/err.txt
    line "$a" "$sched" $out/c2_abl${a}${sched}_$rep.json
  done
done
done
echo "## C4 (pipelined)"
python bench.py --config C4 --steps 24 --warmup 4 --no-cpu-baseline --no-check > $out/c4_product.json 2>$out/err.txt || tail -3 $out/err.txt
line product "" $out/c4_product.json
for a in ${ABLS4:-0 9 41 1}; do
  LIB="--lib $PWD/tools/lib/libgwbp_abl$a.so"
  python bench.py --config C4 --steps 24 --warmup 4 --no-cpu-baseline --no-check --scatter wide $LIB > $out/c4_abl${a}.json 2>$out/err.txt || tail -3 $out/err.txt
  line "$a" "" $out/c4_abl${a}.json
done
} 2>&1 | tee $out/table.txt
