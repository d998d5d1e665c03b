set -x
mkdir -p gpurun_out/r3b
(timeout 900 python -m pytest tests/test_gpu_groups.py -x -q > gpurun_out/r3b/pytest_groups.txt 2>&1; echo rc=$? >> gpurun_out/r3b/pytest_groups.txt)
tail -30 gpurun_out/r3b/pytest_groups.txt
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r3b/bench_C2_groups.json 2> gpurun_out/r3b/bench_C2_groups.err); tail -3 gpurun_out/r3b/bench_C2_groups.err
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --scatter wide > gpurun_out/r3b/bench_C2_wide.json 2> gpurun_out/r3b/bench_C2_wide.err)
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial > gpurun_out/r3b/bench_C2_groups_serial.json 2> gpurun_out/r3b/bench_C2_groups_serial.err)
for f in gpurun_out/r3b/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["launch_ms"], j["config"]["stage_ms"], j["checked"], j["config"]["overflow"])
except Exception as e: print("ERR", e)
PY
done
