mkdir -p gpurun_out/r3i
(timeout 600 python -m pytest tests/test_gpu_groups.py -x -q 2>&1 | tail -2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in "--serial" ""; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3i/ks$mode -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-check --scatter groups $mode > gpurun_out/r3i/bench$mode.json 2>/dev/null
python - "$mode" <<'PY'
import json,sys,csv,glob
m=sys.argv[1]
j=json.load(open("gpurun_out/r3i/bench%s.json"%m))
print("groups", m, "ms/step %.3f"%j["ms_per_step"], {k[:5]:round(v,3) for k,v in j["config"]["stage_ms"].items()})
f=glob.glob("gpurun_out/r3i/ks%s/*/*kernel_stats.csv"%m)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    n=r["Name"].replace("gwbp::(anonymous namespace)::","").replace("void ","").split("(")[0][:34]
    print(f"    {n:36s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
