mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/time_dropin.py 512 > gpurun_out/r3f/time_dropin_C2.txt 2>&1
cat gpurun_out/r3f/time_dropin_C2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f/ks_dropin -- python3 tools/time_dropin.py 512 > /dev/null 2>&1
f=$(find gpurun_out/r3f/ks_dropin -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-160
