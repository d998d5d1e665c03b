#!/bin/bash
# On the GPU box: SQ counters of the front-stage kernels while they run beside the scatter kernel (pipelined bench), per library.
out=${2:-gpurun_out/pmc_front}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in $1; do
  if [ $a = product ]; then LIB=""; else LIB="--lib $PWD/tools/lib/libgwbp_$a.so"; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_$a -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-check $LIB $3 > $out/bench_$a.json 2>/dev/null
  python3 - $out/pmc_$a $a <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("void ","").replace("gwbp::","").replace("(anonymous namespace)::","").split("(")[0]
    if not n.startswith("k_"): continue
    acc[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[n]+=1
print("==",sys.argv[2])
for n in ("k_project","k_radix_scatter","k_blend<1, 4>","k_blend<1>","k_scatter_wide<false>","k_hist"):
    if n in acc:
        c=acc[n]; k=cnt[n]
        print("  %-22s disp %4d  waves %9.0f  wave_cyc %.3g busy %.3g wait_any %.3g wait_inst %.3g valu %.3g gui %.3g"%(n,k,c["SQ_WAVES"]/k,c["SQ_WAVE_CYCLES"]/k,c["SQ_BUSY_CYCLES"]/k,c["SQ_WAIT_ANY"]/k,c["SQ_WAIT_INST_ANY"]/k,c["SQ_INSTS_VALU"]/k,c["GRBM_GUI_ACTIVE"]/k))
PY
done
