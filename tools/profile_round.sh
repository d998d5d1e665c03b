#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r1g'): the default bench line, rocprofv3 kernel stats of the
# same command (pipelined and serial), and the PMC passes that profiles/traffic.json is made from (tools/make_traffic.py).
# Counters go in passes of their own, with --kernel-trace only (gpurun refuses --pmc with the other trace domains).
TAG=${1:-r1x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_pipelined -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench_pipelined_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_serial -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial > $OUT/bench_serial_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial > /dev/null 2>&1
done
find $OUT -name "*.csv" | head -20
