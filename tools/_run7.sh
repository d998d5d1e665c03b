mkdir -p gpurun_out/r3e
(timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest_gpu.txt 2>&1; echo rc=$? >> gpurun_out/r3e/pytest_gpu.txt)
tail -12 gpurun_out/r3e/pytest_gpu.txt
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r3e/bench_C2.json 2>/dev/null
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3e/bench_C2.json"))
print("C2 default: ms/step %.3f"%j["ms_per_step"], j["roofline"]["kernel"], j["config"]["stage_ms"], j["checked"]["ok"])
PY
