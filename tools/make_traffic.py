"""profiles/traffic.json from the PMC passes of tools/profile_round.sh: per-kernel averages of FETCH_SIZE / WRITE_SIZE /
TCC_* per launch, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads; unit KB).  usage: python tools/make_traffic.py gpurun_out/r2a profiles/traffic.json   (every config with pmc_<config>_* directories)"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]


def config_entry(cfg):
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(f"{src}/pmc_{cfg}_*/*/*counter_collection.csv")
    if not files:
        return None
    for f in files:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        for (disp, cname), val in per_dispatch.items():
            raw[names[disp]][cname].append(val)
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in raw.items() if k.startswith("gwbp::")}

    def kb(name, c):
        return avg.get(name, {}).get(c, 0.0) * 1024.0

    # the dominant kernel = the scatter kernel with the most fetched bytes
    cands = [k for k in avg if "k_scatter" in k]
    bl = next((k for k in avg if "k_blend" in k), None)
    # (maps of at most 16 channels: blend and scatter are ONE kernel, gwbp_blend_scatter = k_blend<2>)
    # (round 5: with an encoder the same kernel also encodes, gwbp_blend_scatter_encoded = k_blend<3, 8>)
    # (round 6: its producer / consumer form k_blend<5, 16>; the dino variant's token-space pass k_token_apply<4>)
    fused = (next((k for k in avg if "k_blend<5" in k), None) or next((k for k in avg if "k_blend<3" in k), None)
             or next((k for k in avg if "k_blend<2" in k), None))
    token = next((k for k in avg if "k_token_apply" in k), None)
    sc = token or fused or (max(cands, key=lambda k: avg[k].get("FETCH_SIZE", 0.0)) if cands else bl)
    bl = fused or bl
    if token:
        bl = next((k for k in avg if "k_blend<4" in k), bl)
    a = avg.get(sc, {})
    return {
        "source": f"rocprofv3 --pmc passes of tools/profile_round.sh ({src.rstrip('/').split('/')[-1]}), bench.py --config {cfg} "
                  "--steps 4 --warmup 1 --serial",
        "scatter_kernel": sc,
        "scatter_hbm_bytes_per_launch": 2 * kb(sc, "FETCH_SIZE") + kb(sc, "WRITE_SIZE"),
        "scatter_fetch_bytes_corrected": 2 * kb(sc, "FETCH_SIZE"),
        "scatter_write_bytes": kb(sc, "WRITE_SIZE"),
        "scatter_atomic_requests_64B": a.get("TCC_EA0_ATOMIC_sum", 0.0),
        "scatter_l2_hit_rate": a.get("TCC_HIT_sum", 0.0) / max(1.0, a.get("TCC_HIT_sum", 0.0) + a.get("TCC_MISS_sum", 0.0)),
        "scatter_valu_wave_instructions": a.get("SQ_INSTS_VALU", 0.0),
        "scatter_lds_array_cycles": a.get("SQ_LDS_IDX_ACTIVE", 0.0),
        "blend_hbm_bytes_per_launch": 2 * kb(bl, "FETCH_SIZE") + kb(bl, "WRITE_SIZE"),
        "raw_counters": avg,
    }


import os

out = json.load(open(dst)) if os.path.exists(dst) else {}  # configs without PMC passes in `src` keep their entries
out.update({
    "_about": "HBM traffic per launch from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_*, SQ_* each in its own run "
              "with --kernel-trace only; tools/profile_round.sh), bench.py --steps 4 --warmup 1 --serial, averaged over "
              "launches. Correction per MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the "
              "bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B stores and float atomics. "
              "Counter unit is KB.",
})
for cfg in ("C2", "C4", "C5", "DINO64", "LSEG480"):
    e = config_entry(cfg)
    if e:
        out[cfg] = e
json.dump(out, open(dst, "w"), indent=1)
for cfg in out:
    if cfg != "_about":
        print(cfg, json.dumps({k: v for k, v in out[cfg].items() if k != "raw_counters"}, indent=1))
