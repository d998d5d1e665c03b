#!/bin/bash
# On the GPU box: the dino loop at the other DINOv2 widths (DINO64S / B / G = 384 / 768 / 1536 channels, 64 x 64 tokens, C2 geometry): token space
# with CPU baseline + oracle_check, the pixel-slab kernels (--token-space off), and --serial (k_token_apply alone).  -> profiles/r6_dino_widths.txt
mkdir -p gpurun_out/dinow
for C in DINO64S DINO64B DINO64G; do
  timeout 400 python bench.py --config $C --steps 100 > gpurun_out/dinow/bench_${C}_default.json 2> gpurun_out/dinow/err_$C.txt || tail -3 gpurun_out/dinow/err_$C.txt
  timeout 300 python bench.py --config $C --steps 20 --no-cpu-baseline --token-space off > gpurun_out/dinow/bench_${C}_pixelslab.json 2>> gpurun_out/dinow/err_$C.txt || tail -3 gpurun_out/dinow/err_$C.txt
  timeout 300 python bench.py --config $C --steps 40 --no-cpu-baseline --no-check --serial > gpurun_out/dinow/bench_${C}_serial.json 2>> gpurun_out/dinow/err_$C.txt
done
python - <<'PY'
import json
for C in ("DINO64S","DINO64B","DINO64G"):
    for k in ("default","pixelslab","serial"):
        try:
            j=json.load(open(f"gpurun_out/dinow/bench_{C}_{k}.json"))
            print(C,k,"ms/view %.3f"%j["ms_per_step"], "value %.3e"%j["value"], j["config"].get("stage_ms"), "checked", (j.get("checked") or {}).get("ok"), "oracle", (j.get("oracle_check") or {}).get("ok"), (j.get("oracle_check") or {}).get("F_max_rel_row_err"))
        except Exception as e: print(C,k,"ERR",e)
PY
