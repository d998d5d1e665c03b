#!/bin/bash
# PMC passes of the wide forward render alone (tools/time_render.py), counters of k_render_rows* only.  usage: tools/prof_render.sh [D] [outdir]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=${1:-512}
O=${2:-gpurun_out/r5_render}
mkdir -p $O
python3 tools/time_render.py $D 10 2>&1 | tail -1 > $O/time_$D.txt
n=0
for c in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"; do
  n=$((n+1))
  timeout 240 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_render_rows" --output-format csv -d $O/pmc$n -- python3 tools/time_render.py $D 3 > /dev/null 2>&1
done
cat $O/time_$D.txt
python3 - "$O" <<'P'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_render_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "per launch %.4g" % (sum(v) / len(v)), "launches", len(v))
P
