set -x
mkdir -p gpurun_out/r3a
(timeout 300 tools/ubench_mfma_group 31 0.36 > gpurun_out/r3a/ubench_mfma_group.txt 2>&1; echo rc=$? >> gpurun_out/r3a/ubench_mfma_group.txt)
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest_gpu.txt 2>&1; echo rc=$? >> gpurun_out/r3a/pytest_gpu.txt)
(timeout 300 python bench.py --steps 60 --warmup 5 > gpurun_out/r3a/bench_C2.json 2> gpurun_out/r3a/bench_C2.err)
tail -3 gpurun_out/r3a/pytest_gpu.txt; cat gpurun_out/r3a/ubench_mfma_group.txt; cat gpurun_out/r3a/bench_C2.json | cut -c1-600
