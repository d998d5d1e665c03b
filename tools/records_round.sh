#!/bin/bash
# On the GPU box: the per-round records beside tools/profile_round.sh -- the CLI end to end at BASELINE size, the drop-in's cost,
# the fuzz parity run, a soak line, the one-rank RCCL and the two-ranks-on-one-device bench lines.  usage: tools/records_round.sh <tag>
TAG=${1:-r5b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
for C in C2 C5; do
  echo "== $C"
  for mode in "" "--prune-by-product"; do
    echo "-- python run_backproject.py --synthetic $C $mode"
    t0=$(date +%s.%N)
    python run_backproject.py --synthetic $C --results-dir /tmp/res_$C $mode 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -8
    echo "real $(echo "$(date +%s.%N) - $t0" | bc) s"
    ls -l /tmp/res_$C | tail -3
  done
done
} > $OUT/cli_fullsize.txt 2>&1
python tools/time_dropin.py 512 > $OUT/time_dropin_C2.txt 2>&1
python tools/fuzz_parity.py 3000 5000 > $OUT/fuzz_parity.txt 2>&1
python bench.py --config C2 --steps 2000 --no-cpu-baseline > $OUT/soak_C2.json 2>/dev/null
python bench.py --config C5 --steps 2000 --no-cpu-baseline > $OUT/soak_C5.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --force-dist --no-cpu-baseline > $OUT/bench_C2_force_dist_one_rank_rccl.json 2>/dev/null
python bench.py --gpus 2 --config C2 --steps 8 --warmup 2 --no-cpu-baseline --dist-backend gloo --one-device 2>$OUT/two_ranks.err | grep '^{"metric"' > $OUT/bench_C2_two_ranks_one_device_gloo.json
python bench.py --config C2 > $OUT/bench_C2_default.json 2>/dev/null
tail -3 $OUT/cli_fullsize.txt; tail -3 $OUT/time_dropin_C2.txt; tail -2 $OUT/fuzz_parity.txt
