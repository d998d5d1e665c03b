"""Autograd semantics of the drop-in `rasterization()` beyond the reference's strict forward -> backward -> next view loop
(backproject.py:115-151): several harvest forwards alive at once, several cameras per call, torch.autograd.grad(), hooks,
writes through `.data`.  VERDICT r4 item 7 / ADVICE r4 (high + two medium)."""
import sys

import numpy as np
import pytest
import torch

import gsbp_amd
from gsbp_amd import rasterization
from gsbp_amd import synthetic as syn
from util import rel_row_err, scene_np, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _explicit(dev, cfg, d, v, feats):
    """F_v = scatter of view v's weights over `feats` through the engine (no autograd, no caches)."""
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    view = eng.view(d["vms"][v], d["K"], cfg.width, cfg.height)
    eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
    eng.bin_sort(view)
    eng.blend_weights(view)
    assert not eng.stats()["overflow"]
    F = torch.zeros(cfg.n_gaussians, feats.shape[-1], device=dev)
    eng.scatter(view, feats, F, None)
    return F.cpu().numpy()


@pytest.mark.parametrize("D", [8, 256])
def test_two_harvest_forwards_alive_before_either_backward(dev, D):
    """Two zero tables of one shape, two views, both forwards first, both backwards afterwards: each table's .grad is its own
    view's scatter.  With ONE cached render object handed out twice the first output was re-pointed at the second node and
    every gradient went to the last view."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    args = (d["means"], d["quats"], d["scales"], d["opac"])
    ta = torch.zeros(N, D, device=dev, requires_grad=True)
    tb = torch.zeros(N, D, device=dev, requires_grad=True)
    fa, fb = (syn.make_feature_map(cfg, v, dim=D).to(dev) for v in (0, 1))
    oa, _, _ = rasterization(*args, ta, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
    ob, _, _ = rasterization(*args, tb, d["vms"][1][None], d["K"][None], width=W, height=H, want_meta=False)
    assert oa[0] is not ob[0] and oa[0].grad_fn is not ob[0].grad_fn
    (oa[0] * fa).sum().backward()
    (ob[0] * fb).sum().backward()
    assert rel_row_err(ta.grad.cpu().numpy(), _explicit(dev, cfg, d, 0, fa)) <= TOL
    assert rel_row_err(tb.grad.cpu().numpy(), _explicit(dev, cfg, d, 1, fb)) <= TOL


def test_one_call_with_two_cameras_harvests_both(dev):
    """viewmats [2,4,4] and an all-zero differentiable table: the per-camera renders are separate autograd outputs, the table's
    gradient is the sum of both views' scatters."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    N, W, H, D = cfg.n_gaussians, cfg.width, cfg.height, 8
    t = torch.zeros(N, D, device=dev, requires_grad=True)
    f0, f1 = (syn.make_feature_map(cfg, v, dim=D).to(dev) for v in (0, 1))
    out, alphas, _ = rasterization(d["means"], d["quats"], d["scales"], d["opac"], t, d["vms"][:2], d["K"][None].expand(2, 3, 3),
                                   width=W, height=H, want_meta=False)
    assert out.shape == (2, H, W, D) and alphas.shape == (2, H, W, 1) and float(out.detach().abs().max()) == 0.0
    ((out[0] * f0).sum() + (out[1] * f1).sum()).backward()
    want = _explicit(dev, cfg, d, 0, f0).astype(np.float64) + _explicit(dev, cfg, d, 1, f1)
    assert rel_row_err(t.grad.cpu().numpy(), want) <= TOL
    assert not torch.equal(alphas[0], alphas[1])


def test_autograd_grad_and_hooks_see_a_returned_gradient(dev):
    """The backward may add straight into leaf.grad only under a plain .backward() on an unhooked leaf.  torch.autograd.grad()
    returns the gradient and leaves .grad alone; tensor hooks fire; afterwards the direct path is back and moves .grad's
    version counter."""
    cfg, sc = scene_np("T0")
    d = to_dev(sc, dev)
    N, W, H, D = cfg.n_gaussians, cfg.width, cfg.height, cfg.feat_dim
    args = (d["means"], d["quats"], d["scales"], d["opac"])
    f = syn.make_feature_map(cfg, 0).to(dev)
    want = _explicit(dev, cfg, d, 0, f)
    t = torch.zeros(N, D, device=dev, requires_grad=True)

    def loss():
        out, _, _ = rasterization(*args, t, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
        return (out[0] * f).sum()

    loss().backward()  # first backward: .grad does not exist yet -> returned gradient
    assert rel_row_err(t.grad.cpu().numpy(), want) <= TOL
    g1 = t.grad.clone()
    ver = t.grad._version
    (g,) = torch.autograd.grad(loss(), t)
    assert rel_row_err(g.cpu().numpy(), want) <= TOL
    assert torch.equal(t.grad, g1) and t.grad._version == ver  # .grad untouched
    fired = []
    h = t.register_hook(lambda gr: fired.append(float(gr.abs().sum())) or None)
    loss().backward()
    h.remove()
    assert len(fired) == 1 and fired[0] > 0
    assert rel_row_err(t.grad.cpu().numpy(), 2.0 * want) <= TOL
    ver = t.grad._version
    loss().backward()  # unhooked again: the direct path; it must still look like an in-place update of .grad
    assert t.grad._version > ver
    assert rel_row_err(t.grad.cpu().numpy(), 3.0 * want) <= TOL


def test_a_table_rewritten_through_data_is_not_rendered_as_zero(dev):
    """`.data` writes do not move the version counter that keys the "all zero" verdict: the per-call sample notices a dense
    rewrite and the table is rendered for real."""
    rz = sys.modules["gsbp_amd.rasterization"]
    cfg, sc = scene_np("T0")
    d = to_dev(sc, dev)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    args = (d["means"], d["quats"], d["scales"], d["opac"])
    t = torch.zeros(N, 8, device=dev, requires_grad=True)
    out, _, _ = rasterization(*args, t, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
    assert float(out.detach().abs().max()) == 0.0
    t.data.copy_(torch.rand(N, 8, device=dev))
    out, alpha, _ = rasterization(*args, t, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
    with torch.no_grad():
        ref, ref_alpha, _ = rasterization(*args, t.detach().clone(), d["vms"][0][None], d["K"][None], width=W, height=H,
                                          want_meta=False)
    assert float(out.detach().abs().max()) > 0 and torch.allclose(out.detach(), ref, atol=1e-6)
    t.data.zero_()
    rz.invalidate_zero_table_cache()
    out, _, _ = rasterization(*args, t, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
    assert float(out.detach().abs().max()) == 0.0


def test_harvest_statement_shortcut_equals_the_literal_computation(dev):
    """`(out[0] * feats).sum().backward()` on the render of an all-zero table (backproject.py:127-129) is recognised: no
    product, no sum, no product gradient -- feats itself reaches the scatter.  Same gradient as with the recognition switched
    off (torch computes every statement), as the dino variant's `.mean()` (recognised too: one scaled copy of feats) and a
    partial reduction (NOT recognised: falls back to the literal computation), and as the explicit scatter."""
    rz = sys.modules["gsbp_amd.rasterization"]
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    N, W, H, D = cfg.n_gaussians, cfg.width, cfg.height, 256
    args = (d["means"], d["quats"], d["scales"], d["opac"])
    f = syn.make_feature_map(cfg, 0, dim=D).to(dev)
    want = _explicit(dev, cfg, d, 0, f)

    def run(reduce):
        t = torch.zeros(N, D, device=dev, requires_grad=True)
        out, _, _ = rasterization(*args, t, d["vms"][0][None], d["K"][None], width=W, height=H, want_meta=False)
        prod = out[0] * f
        reduce(prod).backward()
        return t.grad.cpu().numpy(), type(out[0]).__name__, type(prod).__name__

    g_short, t_r, t_p = run(lambda p: p.sum())
    assert (t_r, t_p) == ("_HarvestRender", "_HarvestProduct")
    rz.set_harvest_shortcut(False)
    try:
        g_lit, t_r, t_p = run(lambda p: p.sum())
    finally:
        rz.set_harvest_shortcut(True)
    assert (t_r, t_p) == ("Tensor", "Tensor")
    g_mean, _, _ = run(lambda p: p.mean())
    g_part, _, _ = run(lambda p: p.sum(dim=2).sum())
    assert rel_row_err(g_part, want) <= TOL
    assert rel_row_err(g_short, want) <= TOL and rel_row_err(g_lit, want) <= TOL
    assert rel_row_err(g_short, g_lit) <= 2e-6
    assert rel_row_err(g_mean * (H * W * D), want) <= TOL
