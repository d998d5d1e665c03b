mkdir -p gpurun_out/r3h
for rep in 1 2 3; do
for v in A B; do
  if [ $v = B ]; then export GWBP_LIB=$PWD/tools/lib/libgwbp_B.so; else unset GWBP_LIB; fi
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-check > gpurun_out/r3h/ab_$v$rep.json 2>/dev/null
  python - $v $rep <<'PY'
import json,sys
j=json.load(open("gpurun_out/r3h/ab_%s%s.json"%(sys.argv[1],sys.argv[2])))
print(sys.argv[1], sys.argv[2], "ms/step %.3f"%j["ms_per_step"], {k[:5]:round(v,3) for k,v in j["config"]["stage_ms"].items()})
PY
done; done
