#!/usr/bin/env python3
"""Density of the block-sparse scatter's operand table when its rows are SUPER-RECORDS: one row per (Gaussian, block of
b x b tiles) instead of one per (Gaussian, tile).  (the block-level merge that removes 38 % / 54 % of
the flush atomics.)

For b = 1, 2, 4 on one view of a BASELINE config (CPU oracle): number of super-records (= flushes), groups of 16 after the
8-bit bounding-box sort, mean union size, K-steps (4 pixels each), useful fraction of the dense table, operand-table bytes.

    python tools/superrecord_density.py [C2|C4] [view]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsbp_amd  # noqa: E402,F401
from gsbp_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = synthetic.CONFIGS[name]
    means, quats, scales, opac = [t.numpy() for t in synthetic.activate(synthetic.make_scene(cfg))]
    K = synthetic.intrinsics(cfg).numpy()
    vm = synthetic.make_cameras(cfg)[view].numpy()
    W, H = cfg.width, cfg.height
    proj = orc.project(means, quats, scales, vm, K, W, H)
    bins = orc.bin_sort(proj, W, H)
    gid, pix, w, _ = orc.blend_pairs(proj, bins, opac, W, H)
    gid = gid.astype(np.int64)
    n_pairs = len(gid)
    yy, xx = (pix // W).astype(np.int64), (pix % W).astype(np.int64)
    tw = bins["tile_w"]
    n_rec = len(np.unique(((yy // 16) * tw + xx // 16) * cfg.n_gaussians + gid))
    print(f"{name} view {view}: {n_pairs:,} pairs, {n_rec:,} (Gaussian, tile) records")
    G = 16
    for bs in (1, 2, 4):
        side = 16 * bs
        qw = -(-W // side)
        quad = (yy // side) * qw + xx // side
        key = quad * cfg.n_gaussians + gid
        uk, sr_of_pair = np.unique(key, return_inverse=True)
        n_sr = len(uk)
        sr_quad = uk // cfg.n_gaussians
        py, px = yy % side, xx % side
        y0 = np.full(n_sr, 999); np.minimum.at(y0, sr_of_pair, py)
        x0 = np.full(n_sr, 999); np.minimum.at(x0, sr_of_pair, px)
        y1 = np.full(n_sr, -1); np.maximum.at(y1, sr_of_pair, py)
        x1 = np.full(n_sr, -1); np.maximum.at(x1, sr_of_pair, px)
        cell = 4 * bs  # a 4 x 4 grid of cells per block: the same 8-bit key as k_group_sort's
        okey = ((y0 // cell) * 4 + x0 // cell) * 16 + ((y1 // cell) * 4 + x1 // cell)
        o = np.lexsort((okey, sr_quad))
        q_sorted = sr_quad[o]
        qstart = np.flatnonzero(np.r_[True, q_sorted[1:] != q_sorted[:-1]])
        qcount = np.diff(np.r_[qstart, n_sr])
        slots_per = -(-qcount // G) * G
        slot0 = np.r_[0, np.cumsum(slots_per)[:-1]]
        slot_of_sr = np.empty(n_sr, np.int64)
        slot_of_sr[o] = np.repeat(slot0, qcount) + (np.arange(n_sr) - np.repeat(qstart, qcount))
        n_grp = int(slots_per.sum() // G)
        P = side * side
        U = np.bincount(np.unique((slot_of_sr[sr_of_pair] // G) * P + py * side + px) // P, minlength=n_grp)
        ks = -(-U // 4)
        blocks = -(-ks // 4)
        print(f"  {bs}x{bs} tile blocks: {n_sr:,} super-records = {n_sr / n_rec:.3f} of the flushes, {n_grp:,} groups "
              f"({n_grp / len(qstart):.1f} per block), mean union {U.mean():.0f} px, {ks.sum():,} K-steps, "
              f"{n_pairs / (ks.sum() * 4 * G):.3f} of the table non-zero, operand table {blocks.sum() * 1040 / 1e9:.2f} GB")


if __name__ == "__main__":
    main()
