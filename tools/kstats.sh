#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics of the pipelined bench for the listed library builds (tools/lib/libgwbp_<name>.so;
# "product" = the in-tree library).  usage: tools/kstats.sh "<names>" <out dir> [bench args]
out=${2:-gpurun_out/kstats}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in $1; do
  if [ $a = product ]; then LIB=""; else LIB="--lib $PWD/tools/lib/libgwbp_$a.so"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$a -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-check $LIB $3 > $out/bench_$a.json 2>/dev/null
  f=$(find $out/ks_$a -name "*kernel_stats.csv" | head -1)
  echo "== $a  $(python3 -c "import json;print(round(json.load(open('$out/bench_$a.json'))['ms_per_step'],3))") ms/step"
  python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"].replace("void ","").replace("gwbp::","").replace("(anonymous namespace)::","").split("(")[0]
    print("   %-34s calls %5s avg %9.1f us  total %8.2f ms"%(n[:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
