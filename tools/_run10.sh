mkdir -p gpurun_out/r3g
(timeout 900 python -m pytest tests/test_gpu_groups.py -x -q > gpurun_out/r3g/pytest_groups.txt 2>&1; echo rc=$? >> gpurun_out/r3g/pytest_groups.txt)
tail -4 gpurun_out/r3g/pytest_groups.txt
for mode in "--serial" ""; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-check --scatter groups $mode > gpurun_out/r3g/bench_groups$mode.json 2>/dev/null
python - "$mode" <<'PY'
import json,sys
j=json.load(open("gpurun_out/r3g/bench_groups%s.json"%sys.argv[1]))
print("groups", sys.argv[1], "ms/step %.3f"%j["ms_per_step"], j["config"]["stage_ms"])
PY
done
export GWBP_LIB=$PWD/tools/lib/libgwbp_profile.so GWBP_ALLOW_PROFILE=1
for ab in 0 1 2 3 7; do
  GWBP_ABLATE=$ab python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check --serial --scatter groups > gpurun_out/r3g/ab${ab}.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("gpurun_out/r3g/ab${ab}.json"))
print("ablate=$ab serial ms/step %.3f scatter %.3f"%(j["ms_per_step"], j["config"]["stage_ms"]["scatter"]))
PY
done
