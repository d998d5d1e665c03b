#!/usr/bin/env python3
"""How dense would a block-sparse (matrix-core) formulation of the scatter be?  

The scatter computes F[g,:] += sum_p w[g,p] * feats[p,:] per tile: a sparse (Gaussians x 256 pixels) matrix times a dense
(256 pixels x D) slab.  Today every non-zero w costs one LDS row read + D FMAs on the vector ALUs.  On the matrix cores
(fp32 MFMA = the fp32 vector rate on gfx950) a group of G Gaussians shares each pixel read, but every (group, pixel-block)
step costs G x P products whether the weights are zero or not.  This script takes the CPU oracle's weight list of one
view of a BASELINE config and reports, for several blockings and record orderings, the fraction of block products that
are useful (= non-zero weights / products issued).  The matrix-core time per view is then (sparse FMA floor) / fraction.

CPU only (needs oracle/): python tools/block_density.py [config] [view]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsbp_amd  # noqa: E402,F401
from gsbp_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def morton(y, x):
    out = np.zeros_like(y, dtype=np.int64)
    for b in range(4):
        out |= ((x >> b) & 1).astype(np.int64) << (2 * b)
        out |= ((y >> b) & 1).astype(np.int64) << (2 * b + 1)
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = synthetic.CONFIGS[name]
    scene = synthetic.make_scene(cfg)
    means, quats, scales, opac = [t.numpy() for t in synthetic.activate(scene)]
    K = synthetic.intrinsics(cfg).numpy()
    vm = synthetic.make_cameras(cfg)[view].numpy()
    W, H = cfg.width, cfg.height
    t0 = time.time()
    proj = orc.project(means, quats, scales, vm, K, W, H)
    bins = orc.bin_sort(proj, W, H)
    gid, pix, w, _ = orc.blend_pairs(proj, bins, opac, W, H)
    print(f"{name} view {view}: {len(gid):,} pairs, {bins['n_isect']:,} intersections ({time.time() - t0:.0f} s)")
    tw = bins["tile_w"]
    py, px = pix // W, pix % W
    tile = (py // 16) * tw + px // 16
    local = ((py % 16) * 16 + px % 16).astype(np.int32)
    key = tile.astype(np.int64) * cfg.n_gaussians + gid
    order = np.argsort(key, kind="stable")
    key, local = key[order], local[order]
    first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    n_rec = len(first)
    rec_of_pair = np.cumsum(np.r_[True, key[1:] != key[:-1]]) - 1
    counts = np.diff(np.r_[first, len(key)])
    rec_tile = (key[first] // cfg.n_gaussians).astype(np.int64)
    rec_gid = (key[first] % cfg.n_gaussians).astype(np.int64)
    print(f"records with weight: {n_rec:,}; pairs per record: mean {counts.mean():.1f}, median {np.median(counts):.0f}, "
          f"quartiles {np.percentile(counts, 25):.0f}/{np.percentile(counts, 75):.0f}, "
          f">=128: {(counts >= 128).mean():.3f}, <=8: {(counts <= 8).mean():.3f}")
    for lo, hi in ((1, 8), (9, 32), (33, 96), (97, 192), (193, 256)):
        sel = (counts >= lo) & (counts <= hi)
        print(f"   records with {lo:3d}..{hi:3d} pairs: {sel.mean():.3f} of records, {counts[sel].sum() / counts.sum():.3f} of pairs")

    mask = np.zeros((n_rec + 1, 256), np.uint8)  # last row = padding
    mask[rec_of_pair, local] = 1
    ly, lx = local // 16, local % 16
    cy = np.bincount(rec_of_pair, ly, n_rec) / counts
    cx = np.bincount(rec_of_pair, lx, n_rec) / counts

    # depth rank of a record inside its tile = its position in the tile's sorted list
    offs = bins["tile_offsets"]
    flat = bins["flatten_ids"]
    isect_tile = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
    isect_key = isect_tile.astype(np.int64) * cfg.n_gaussians + flat
    rank_all = np.arange(len(flat)) - offs[isect_tile]
    sorter = np.argsort(isect_key, kind="stable")
    pos = np.searchsorted(isect_key[sorter], key[first])
    depth_rank = rank_all[sorter[pos]]

    size_class = np.minimum(np.log2(counts).astype(np.int64), 7)
    orderings = {
        "depth order (as blended)": depth_rank.astype(np.int64),
        "centre, Morton": morton(cy.astype(np.int64), cx.astype(np.int64)),
        "size class, then centre": size_class * 256 + morton(cy.astype(np.int64), cx.astype(np.int64)),
    }
    pix_blocks = {
        "1 px": np.arange(256),
        "4x1 px": (np.arange(256) // 4),
        "2x2 px": ((np.arange(256) // 32) * 8 + (np.arange(256) % 16) // 2),
    }
    total = counts.sum()
    print(f"\nuseful fraction of block products (1.0 = today's sparse work; matrix-core time = {total * cfg.feat_dim * 2 / 157.3e12 * 1e3:.2f} ms / fraction)")
    for oname, okey in orderings.items():
        o = np.lexsort((okey, rec_tile))
        t_sorted = rec_tile[o]
        tstart = np.flatnonzero(np.r_[True, t_sorted[1:] != t_sorted[:-1]])
        tcount = np.diff(np.r_[tstart, n_rec])
        for G in (4, 16, 32):
            slots_per_tile = -(-tcount // G) * G
            slot0 = np.r_[0, np.cumsum(slots_per_tile)[:-1]]
            slots = np.full(int(slots_per_tile.sum()), n_rec, np.int64)
            within = np.arange(n_rec) - np.repeat(tstart, tcount)
            slots[np.repeat(slot0, tcount) + within] = o
            grouped = mask[slots].reshape(-1, G, 256)
            union = grouped.any(1)  # [groups, 256]
            row = []
            for pname, pb in pix_blocks.items():
                nb = pb.max() + 1
                P = 256 // nb
                blk = np.zeros((union.shape[0], nb), bool)
                for b in range(nb):
                    blk[:, b] = union[:, pb == b].any(1)
                steps = int(blk.sum())
                row.append(f"{pname}: {total / (steps * G * P):.3f}")
            print(f"  {oname:28s} G={G:2d}  " + "   ".join(row))


if __name__ == "__main__":
    main()
