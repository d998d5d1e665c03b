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
