#!/usr/bin/env python3
"""Would an on-chip accumulator cache merge enough flushes?  

The scatter kernels flush one D-wide row of partial sums per (Gaussian, tile) record with fp32 atomics; the atomic unit's
op rate is the floor of the whole design (C4 runs at 0.9 of it).  Proposal: a persistent workgroup walks the tiles of a
2x2 / 4x4 block back to back (Morton order) and keeps a small direct-mapped cache of accumulator rows in the LDS the slab
leaves free (<= 32 KB: 64 rows of a 128-channel chunk, 32 rows of a 256-channel chunk), keyed by Gaussian id: a record
whose Gaussian sits in the cache adds into LDS instead of flushing, an eviction flushes.

This script replays the record stream of one view of a BASELINE config (CPU oracle) through that cache and reports the
fraction of flushes it would save, for several cache sizes / replacement policies / block shapes, next to the upper bound
(every Gaussian flushed once per block of tiles, i.e. an infinite cache).

    python tools/flush_cache_sim.py [C2|C4] [view]
"""
import os
import sys
import time
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsbp_amd  # noqa: E402,F401
from gsbp_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def records(name, view):
    cfg = synthetic.CONFIGS[name]
    means, quats, scales, opac = [t.numpy() for t in synthetic.activate(synthetic.make_scene(cfg))]
    K = synthetic.intrinsics(cfg).numpy()
    vm = synthetic.make_cameras(cfg)[view].numpy()
    W, H = cfg.width, cfg.height
    proj = orc.project(means, quats, scales, vm, K, W, H)
    bins = orc.bin_sort(proj, W, H)
    gid, pix, w, _ = orc.blend_pairs(proj, bins, opac, W, H)
    tw = bins["tile_w"]
    tile = ((pix // W) // 16) * tw + (pix % W) // 16
    key = np.unique(tile.astype(np.int64) * cfg.n_gaussians + gid)  # one per contributing (tile, Gaussian)
    return (key // cfg.n_gaussians).astype(np.int64), (key % cfg.n_gaussians).astype(np.int64), tw, -(-H // 16), len(gid)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    rt, rg, tw, th, n_pairs = records(name, view)
    n_rec = len(rt)
    print(f"{name} view {view}: {n_rec:,} records, {n_pairs:,} pairs, {len(np.unique(rg)):,} Gaussians with weight "
          f"({n_rec / len(np.unique(rg)):.2f} records per Gaussian) [{time.time() - t0:.0f} s]")
    ty, tx = rt // tw, rt % tw
    for bs in (2, 4):
        blk = (ty // bs) * (-(-tw // bs)) + tx // bs
        once = len(np.unique(blk * (rg.max() + 1) + rg))
        print(f"  {bs}x{bs} tile blocks, infinite cache (one flush per Gaussian and block): {1 - once / n_rec:.3f} of the flushes saved")
        # replay: blocks in order, tiles of a block in Morton order, records of a tile in (sorted-by-gid) list order
        sub = (ty % bs) * bs + tx % bs
        order = np.lexsort((rg, sub, blk))
        b_s, g_s = blk[order], rg[order]
        starts = np.flatnonzero(np.r_[True, b_s[1:] != b_s[:-1]])
        ends = np.r_[starts[1:], n_rec]
        sample = np.linspace(0, len(starts) - 1, min(len(starts), 400)).astype(int)  # 400 blocks are plenty
        for rows in (32, 64, 128, 256, 1024):
            hits_dm = hits_lru = total = 0
            for bi in sample:
                gs = g_s[starts[bi]:ends[bi]]
                total += len(gs)
                dm = np.full(rows, -1, np.int64)
                lru = OrderedDict()
                for g in gs.tolist():
                    s = g % rows
                    if dm[s] == g:
                        hits_dm += 1
                    dm[s] = g
                    if g in lru:
                        hits_lru += 1
                        lru.move_to_end(g)
                    else:
                        lru[g] = 1
                        if len(lru) > rows:
                            lru.popitem(last=False)
            print(f"     cache of {rows:4d} rows: direct-mapped saves {hits_dm / total:.3f}, LRU saves {hits_lru / total:.3f} of the flushes")


if __name__ == "__main__":
    main()
