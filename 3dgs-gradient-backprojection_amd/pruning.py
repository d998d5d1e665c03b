"""Pruning by gradients and its check: counterparts of utils.prune_by_gradients (utils.py:222-271) and
utils.test_proper_pruning (utils.py:292-360) on the HIP path.

The reference rasterises every view with the DC colours (no SH, utils.py:238-249), back-propagates the pseudo-loss
((out.detach() + 1 - out) ** 2).mean() whose gradient w.r.t. `out` is the constant -2 / (H W 3), accumulates
colors.grad[:, 0].norm() per Gaussian and keeps the Gaussians whose sum is > 0 (utils.py:251-257).  That gradient is
-2/(3HW) * sum_p w_g(p) in every channel, so the mask is exactly  sum_v d_v[g] > 0  -- the denominator the fused
back-projection accumulates anyway.  `prune_by_gradients` here gets it from one blend per view (no scatter at all: the
blend leaves every record's weight sum in its header and gwbp_accumulate_d adds them).  The reference's own loop run through
the drop-in rasterization() + autograd gives the same mask (tests/util.py::gradient_mask_literal, tests/test_gpu_pruning.py).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .engine import Engine
from .rasterization import rasterization
from .synthetic import view_shard

_PER_GAUSSIAN = ("means", "features_dc", "features_rest", "scaling", "rotation", "opacity", "features")


def _activated(splats: Dict[str, torch.Tensor]):
    """utils.py:231-233 / backproject.py:55-57: sigmoid opacities, exp scales, raw quaternions."""
    return (splats["means"], splats["rotation"], torch.exp(splats["scaling"]), torch.sigmoid(splats["opacity"]))


def _group():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def gradient_mask(splats: Dict[str, torch.Tensor], viewmats: torch.Tensor, K: torch.Tensor, width: int,
                  height: int) -> torch.Tensor:
    """bool[N]: Gaussians that receive weight in at least one of the views (utils.py:236-257).

    Under a process group the views are sharded r, r + R, ... like the field build and the per-rank weight sums are
    all-reduced: every rank returns the SAME mask (an all-reduce hands every rank the same bits), so the Gaussian slices and
    the shapes of the collectives that follow agree by construction."""
    dist, rank, world = _group()
    means, quats, scales, opac = _activated(splats)
    n, dev = means.shape[0], means.device
    width, height = int(width), int(height)  # utils.py:247-248 passes 0-d tensors
    eng = Engine(n, width, height, device=dev, tight_binning=True)
    eng.set_narrow_scatter(False)  # the blend then leaves every record's weight sum in its header
    d = torch.zeros(n, device=dev)
    vm_host, K_host = viewmats.detach().cpu(), K.detach().cpu()
    for v in view_shard(viewmats.shape[0], rank, world):
        view = eng.view(vm_host[v], K_host, width, height)
        while True:
            eng.project(view, means, quats, scales, opac)
            eng.bin_sort(view)
            eng.blend_weights(view)
            st = eng.stats()  # one host sync per view: this is a one-off pre-pass, not the hot loop
            if not st["overflow"]:
                break
            eng.grow(st)
        eng.accumulate_d(view, d)  # after the overflow check: a retried view must not be counted twice
    if dist is not None:
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
    return d > 0


def prune_by_gradients(splats: Dict[str, torch.Tensor], viewmats: torch.Tensor, K: torch.Tensor, width: int,
                       height: int) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """(pruned copy of the splats dict, mask): utils.py:222-271 with the cameras passed explicitly instead of through
    splats["colmap_project"]."""
    mask = gradient_mask(splats, viewmats, K, width, height)
    out = dict(splats)
    for k in _PER_GAUSSIAN:
        if k in out:
            out[k] = out[k][mask]
    return out, mask


def check_proper_pruning(splats: Dict[str, torch.Tensor], pruned: Dict[str, torch.Tensor], viewmats: torch.Tensor,
                         K: torch.Tensor, width: int, height: int) -> Dict[str, float]:
    """utils.test_proper_pruning (utils.py:292-360): SH-degree-3 render of every view before and after pruning through
    the drop-in rasterization(); asserts max |difference| < 1 / (255 * 2) like utils.py:353-355.  Under a process group each rank
    renders its shard of the views; the maximum and the total are reduced, every rank asserts on the same numbers."""
    dist, rank, world = _group()

    def cols(s):
        return torch.cat([s["features_dc"], s["features_rest"]], dim=1)

    a, b = _activated(splats), _activated(pruned)
    total, worst = 0.0, 0.0
    with torch.no_grad():
        for v in view_shard(viewmats.shape[0], rank, world):
            kw = dict(viewmats=viewmats[v][None], Ks=K[None], sh_degree=3, width=int(width), height=int(height),
                      want_meta=False)
            out, _, _ = rasterization(*a, cols(splats), **kw)
            out_p, _, _ = rasterization(*b, cols(pruned), **kw)
            diff = (out - out_p).abs()
            total += float(diff.sum())
            worst = max(worst, float(diff.max()))
    if dist is not None:
        t = torch.tensor([worst, total], dtype=torch.float64, device=viewmats.device)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        worst, total = float(t[0]), float(t[1])
    n0, n1 = splats["means"].shape[0], pruned["means"].shape[0]
    assert worst < 1 / (255 * 2), "Max pixel error should be less than 1/(255*2), safety margin"
    return {"percentage_pruned": 100.0 * (n0 - n1) / max(n0, 1), "max_pixel_error": worst, "total_pixel_error": total}
