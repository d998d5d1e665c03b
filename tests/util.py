"""Shared helpers for the parity tests (scene -> numpy / device tensors)."""
import numpy as np
import torch

import gsbp_amd  # noqa: F401
from gsbp_amd import synthetic as syn


def scene_np(name, **over):
    cfg = syn.CONFIGS[name]
    if over:
        cfg = syn.Config(**{**cfg.__dict__, **over})
    means, quats, scales, opac = syn.activate(syn.make_scene(cfg))
    return cfg, dict(means=means, quats=quats, scales=scales, opac=opac, K=syn.intrinsics(cfg),
                     vms=syn.make_cameras(cfg))


def to_dev(sc, dev):
    return {k: v.to(dev) for k, v in sc.items()}


def npy(sc):
    return {k: v.numpy() for k, v in sc.items()}


def rel_row_err(a: np.ndarray, ref: np.ndarray, floor=1e-30):
    """max over rows of ||a - ref|| / max(||ref||, floor-scaled)   (SURVEY.md 8(d) parity check)."""
    num = np.linalg.norm(a.astype(np.float64) - ref.astype(np.float64), axis=-1)
    den = np.linalg.norm(ref.astype(np.float64), axis=-1)
    scale = max(den.max(), floor)
    return float((num / np.maximum(den, 1e-6 * scale)).max())


def sort_pairs(gid, pix, w):
    key = gid.astype(np.int64) * (1 << 32) + pix.astype(np.int64)
    o = np.argsort(key, kind="stable")
    return key[o], w[o]


def gradient_mask_literal(splats, viewmats, K, width, height):
    """TEST SUPPORT: the reference's own pruning loop (utils.py:236-257) run through the drop-in rasterization() +
    autograd -- DC colours, pseudo-loss ((out.detach() + 1 - out) ** 2).mean(), colors.grad[:, 0].norm() summed over the
    views, mask = sum > 0.  The product (gsbp_amd.pruning.gradient_mask) gets the same mask from one blend per view."""
    from gsbp_amd import rasterization
    means, quats = splats["means"], splats["rotation"]
    scales, opac = torch.exp(splats["scaling"]), torch.sigmoid(splats["opacity"])
    colors = torch.cat([splats["features_dc"], splats["features_rest"]], dim=1).detach().clone()
    colors.requires_grad = True
    grads = torch.zeros(means.shape[0], device=means.device)
    for v in range(viewmats.shape[0]):
        out, _, _ = rasterization(means, quats, scales, opac, colors[:, 0, :], viewmats=viewmats[v][None],
                                  Ks=K[None], width=width, height=height, want_meta=False)
        loss = ((out.detach() + 1 - out) ** 2).mean()
        loss.backward()
        grads += colors.grad[:, 0].norm(dim=[1])
        colors.grad.zero_()
    return grads > 0


def capture_report(cap, out, F, d, radii=None, means2d=None, conics=None, depths=None, n_pairs=None):
    """Compare a result (oracle or HIP) with a capture of REAL gsplat 1.4.0 output (tools/capture_gsplat_fixture.py).
    Returns a dict of per-quantity errors and the census of rows over the north_star tolerance.  Rows whose pairs sit
    exactly on the alpha >= 1/255 / T' <= 1e-4 cuts are EXPECTED to exceed 1e-4 when gsplat's __expf differs from the
    deterministic polynomial by an ulp or two (profiles/r2_sensitivity_C1.json: 2-3 rows of 10 000 at 2 ulp), so the
    caller asserts on the bulk (median, 99th percentile, number of rows), not on the maximum."""
    rep = {"gsplat_version": str(cap["gsplat_version"]) if "gsplat_version" in cap else "?"}
    Fc, dc, oc = cap["F"].astype(np.float64), cap["d"].astype(np.float64), cap["out"].astype(np.float64)
    fn = np.linalg.norm(Fc, axis=1)
    scale = np.maximum(fn, 1e-6 * max(fn.max(), 1e-30))
    row_f = np.linalg.norm(np.asarray(F, np.float64) - Fc, axis=1) / scale
    row_d = np.abs(np.asarray(d, np.float64) - dc) / np.maximum(dc, 1e-6 * max(dc.max(), 1e-30))
    row_o = np.abs(np.asarray(out, np.float64) - oc).max(axis=1)
    for name, r in (("F", row_f), ("d", row_d), ("out", row_o)):
        rep[name] = {"max": float(r.max()), "p99": float(np.percentile(r, 99)), "median": float(np.median(r)),
                     "rows_over_1e-4": int((r > 1e-4).sum()), "rows": int(r.size)}
    # forward meta of view 0 (gsplat packs visible Gaussians: index by gaussian_ids when present)
    ids = cap["v0_gaussian_ids"] if "v0_gaussian_ids" in cap else None
    for name, mine in (("radii", radii), ("means2d", means2d), ("conics", conics), ("depths", depths)):
        key = "v0_" + name
        if mine is None or key not in cap:
            continue
        theirs = np.asarray(cap[key])
        theirs = theirs.reshape(-1, *theirs.shape[2:]) if theirs.ndim > np.asarray(mine).ndim else theirs  # [C, N, ..] -> [N, ..]
        sel = np.asarray(mine)[ids] if ids is not None and theirs.shape[0] == len(ids) else np.asarray(mine)
        if theirs.shape != sel.shape:
            rep[name] = {"shape_mismatch": [list(theirs.shape), list(sel.shape)]}
            continue
        if name == "radii":
            vis = theirs > 0
            rep[name] = {"visible_equal": bool(np.array_equal(vis, sel > 0)), "n_diff": int((theirs != sel).sum())}
        else:
            vis = np.isfinite(theirs).reshape(theirs.shape[0], -1).all(1)
            err = np.abs(theirs[vis].astype(np.float64) - sel[vis]) / np.maximum(np.abs(theirs[vis]), 1e-6)
            rep[name] = {"max_rel": float(err.max()) if err.size else 0.0}
    return rep


def capture_tool():
    """tools/capture_gsplat_fixture.py as a module: its CASES table and case_inputs() are shared with the consumer tests."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "capture_gsplat_fixture.py")
    spec = importlib.util.spec_from_file_location("capture_gsplat_fixture", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
