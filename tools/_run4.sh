set -x
mkdir -p gpurun_out/r3c
(timeout 900 python -m pytest tests/test_gpu_groups.py -x -q > gpurun_out/r3c/pytest_groups.txt 2>&1; echo rc=$? >> gpurun_out/r3c/pytest_groups.txt)
tail -5 gpurun_out/r3c/pytest_groups.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/ks_serial -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --serial --no-check > gpurun_out/r3c/bench_serial.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/ks_pipe -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check > gpurun_out/r3c/bench_pipe.json 2> /dev/null
for f in $(find gpurun_out/r3c -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -d, -f1-5 | cut -c1-150; done
