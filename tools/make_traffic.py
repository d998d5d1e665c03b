"""profiles/traffic.json from the PMC passes of tools/profile_round.sh: per-kernel averages of FETCH_SIZE / WRITE_SIZE /
TCC_* per launch, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads; unit KB).  usage: python tools/make_traffic.py gpurun_out/r1g profiles/traffic.json"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
raw = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/pmc_*/*/*counter_collection.csv"):
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


sc = next((k for k in avg if "k_scatter_wide" in k), None) or next((k for k in avg if "k_scatter_full" in k), None)
bl = next((k for k in avg if "k_blend" in k), None)
out = {
    "_about": "HBM traffic per launch from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_* each in its own run with "
              "--kernel-trace only; tools/profile_round.sh), bench.py --steps 4 --warmup 1 --serial on C2, averaged over "
              "launches. Correction per MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the "
              "bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B stores and float atomics. "
              "Counter unit is KB.",
    "C2": {
        "scatter_kernel": sc,
        "scatter_hbm_bytes_per_launch": 2 * kb(sc, "FETCH_SIZE") + kb(sc, "WRITE_SIZE"),
        "scatter_fetch_bytes_corrected": 2 * kb(sc, "FETCH_SIZE"),
        "scatter_write_bytes": kb(sc, "WRITE_SIZE"),
        "scatter_atomic_requests_64B": avg.get(sc, {}).get("TCC_EA0_ATOMIC_sum", 0.0),
        "scatter_l2_hit_rate": (avg.get(sc, {}).get("TCC_HIT_sum", 0.0) /
                                max(1.0, avg.get(sc, {}).get("TCC_HIT_sum", 0.0) + avg.get(sc, {}).get("TCC_MISS_sum", 0.0))),
        "blend_hbm_bytes_per_launch": 2 * kb(bl, "FETCH_SIZE") + kb(bl, "WRITE_SIZE"),
        "raw_counters": avg,
    },
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out["C2"].items() if k != "raw_counters"}, indent=1))
