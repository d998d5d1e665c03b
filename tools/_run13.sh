for v in A w5 w6; do
  if [ $v = A ]; then unset GWBP_LIB; else export GWBP_LIB=$PWD/tools/lib/libgwbp_$v.so; fi
  for mode in "" "--serial"; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --scatter groups $mode > /tmp/b.json 2>/dev/null
  python - $v "$mode" <<'PY'
import json,sys
j=json.load(open("/tmp/b.json"))
print(sys.argv[1], sys.argv[2], "ms/step %.3f"%j["ms_per_step"], {k[:5]:round(v,3) for k,v in j["config"]["stage_ms"].items()}, j["checked"]["ok"])
PY
done; done
