#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r2a'): per BASELINE config the default bench line, rocprofv3
# kernel stats of the same command, and the PMC passes that profiles/traffic.json is made from (tools/make_traffic.py).
# Counters go in passes of their own, with --kernel-trace only (gpurun refuses --pmc with the other trace domains).
TAG=${1:-r2x}
CONFIGS=${2:-"C2 C4 C5 C1 DINO64 LSEG480"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in $CONFIGS; do
  case $C in C2) STEPS=200;; C4) STEPS=60;; DINO64|LSEG480) STEPS=100;; *) STEPS=200;; esac
  if [ -z "$SKIP_BENCH" ]; then
    python3 bench.py --config $C --steps $STEPS > $OUT/bench_${C}_default.json 2> $OUT/bench_${C}_default.err
    tail -c 400 $OUT/bench_${C}_default.json; echo
  fi
  [ -n "$ONLY_BENCH" ] && continue
  case $C in C4) PS=20;; *) PS=60;; esac
  # the serial PMC runs must use the scatter kernel the pipelined driver settles on (C4: short records -> 128-channel)
  case $C in C4) SK="--scatter narrow";; *) SK="";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_${C}_pipelined -- python3 bench.py --config $C --steps $PS --warmup 5 --no-cpu-baseline --no-check > $OUT/bench_${C}_pipelined_under_rocprof.json 2>/dev/null
  if [ $C = C2 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_${C}_serial -- python3 bench.py --config $C --steps 40 --warmup 5 --no-cpu-baseline --no-check --serial > $OUT/bench_${C}_serial_under_rocprof.json 2>/dev/null
  fi
  if [ $C != C1 ]; then
    for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
      n=$(echo $c | cut -d' ' -f1)
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${C}_$n -- python3 bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --no-check --serial $SK > /dev/null 2>&1
    done
  fi
done
find $OUT -name "*stats.csv" | head -20
