#!/usr/bin/env python3
"""Generates tests/golden/g0.npz: inputs + expected outputs of the hot path on a tiny seeded scene.

The reference itself cannot run here (gsplat 1.4.0 is CUDA-only and not vendored), so these
vectors come from the CPU oracle (oracle/gwbp_oracle.c), which tests/test_oracle.py cross-checks against an
independent float64 formulation, a torch-autograd restatement of the backproject.py loop and closed-form cases.
They pin the arithmetic contract: any drift of the oracle OR of the HIP kernels shows up as a mismatch.

    python tests/golden/make_golden.py        # rewrites g0.npz (only when the contract changes on purpose)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import gsbp_amd  # noqa: E402,F401
from gsbp_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CFG = syn.Config("G0", 256, 2, 64, 48, 8, 0.08, False)


def main():
    means, quats, scales, opac = [t.numpy() for t in syn.activate(syn.make_scene(CFG))]
    K = syn.intrinsics(CFG).numpy()
    vms = syn.make_cameras(CFG).numpy()
    feats = np.stack([syn.make_feature_map(CFG, v).numpy() for v in range(CFG.n_views)])
    out, F, d, stats = orc.backproject_oracle(means, quats, scales, opac, vms, K, CFG.width, CFG.height,
                                              lambda v: feats[v], CFG.feat_dim)
    proj = orc.project(means, quats, scales, vms[0], K, CFG.width, CFG.height)
    bins = orc.bin_sort(proj, CFG.width, CFG.height)
    gid, pix, w, alphas = orc.blend_pairs(proj, bins, opac, CFG.width, CFG.height, want_alphas=True)
    xs = np.linspace(-20, 0, 41, dtype=np.float32)
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "g0.npz"),
        means=means, quats=quats, scales=scales, opac=opac, K=K, vms=vms, feats=feats,
        out=out, F=F.astype(np.float64), d=d.astype(np.float64),
        n_pairs=np.array([s["n_pairs"] for s in stats]), n_isect=np.array([s["n_isect"] for s in stats]),
        v0_radii=proj["radii"], v0_means2d=proj["means2d"], v0_conics=proj["conics"], v0_depths=proj["depths"],
        v0_isect_ids=bins["isect_ids"], v0_flatten_ids=bins["flatten_ids"], v0_tile_offsets=bins["tile_offsets"],
        v0_pair_gid=gid, v0_pair_pix=pix, v0_pair_w=w, v0_alphas=alphas,
        exp_x=xs, exp_y=np.array([orc.exp_neg(float(x)) for x in xs], np.float32))
    print("wrote g0.npz:", {k: v for k, v in zip(("n_pairs", "n_isect"), (stats[0]["n_pairs"], stats[0]["n_isect"]))})


if __name__ == "__main__":
    main()
