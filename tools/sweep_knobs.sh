#!/bin/bash
# On the GPU box: pipeline knobs of bench.py (workspaces in flight, wave priority of the front kernels, stream priority).
# usage: tools/sweep_knobs.sh "<args 1>" "<args 2>" ...
for args in "$@"; do
  python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-check $args > /tmp/b.json 2>/dev/null
  python - "$args" <<'PY'
import json,sys
j=json.load(open("/tmp/b.json")); sm=j["config"]["stage_ms"]
print("%-45s ms/step %.3f  %s"%(sys.argv[1], j["ms_per_step"], {k[:8]:round(v,3) for k,v in sm.items()}))
PY
done
