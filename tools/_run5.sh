mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-check > gpurun_out/r3d/bench_pipe_prio.json 2>/dev/null
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3d/bench_pipe_prio.json"))
print("pipelined + front priority: ms/step %.3f"%j["ms_per_step"], j["config"]["stage_ms"])
PY
export GWBP_LIB=$PWD/tools/lib/libgwbp_profile.so GWBP_ALLOW_PROFILE=1
for ab in 0 1 2 4 3 6 7; do
  GWBP_ABLATE=$ab python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check --serial > gpurun_out/r3d/ab${ab}.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("gpurun_out/r3d/ab${ab}.json"))
print("ablate=$ab serial ms/step %.3f scatter %.3f"%(j["ms_per_step"], j["config"]["stage_ms"]["scatter"]))
PY
done
unset GWBP_LIB GWBP_ALLOW_PROFILE
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_LDS_[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/r3d/counters.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3d/pmc_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --serial > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/r3d/pmc_*/*/*counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for name in ("k_scatter_mfma","k_pack","k_blend","k_group_sort"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name,d in acc.items():
        print(name, {c: "%.3e"%(sum(v)/len(v)) for c,v in d.items()})
PY
cat gpurun_out/r3d/counters.txt
