"""Full-size (BASELINE.json configs[1]: 1M Gaussians, 1600x1060) GPU checks through size-independent properties,
plus one oracle comparison at full geometry.  One view each; ~1 minute on the GPU box."""
import numpy as np
import pytest
import torch

from util import rel_row_err

import gsbp_amd
from gsbp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(dev):
    cfg = syn.CONFIGS["C2"]
    means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    vms, K = syn.make_cameras(cfg, n_views=2), syn.intrinsics(cfg)
    return cfg, eng, (means, quats, scales, opac), vms, K


def _run(eng, cfg, g, vm, K, feats, dev):
    F = torch.zeros(cfg.n_gaussians, feats.shape[2], device=dev)
    d = torch.zeros(cfg.n_gaussians, device=dev)
    eng.backproject_view(eng.view(vm, K, cfg.width, cfg.height), *g, feats, F, d)
    st = eng.stats()
    assert st["overflow"] == 0
    return F, d, st


def test_conservation_and_constant_feature_invariant(c2, dev):
    """sum_g d[g] == sum_p (1 - T_p)  (w = T - T' telescopes per pixel), and a channel-constant map gives
    F[g, c] == d[g] for every channel (demo_affordance_transfer.py:383-386)."""
    cfg, eng, g, vms, K = c2
    D = 128
    ones = torch.ones(cfg.height, cfg.width, D, device=dev)
    F, d, st = _run(eng, cfg, g, vms[0], K, ones, dev)
    view = eng.view(vms[0], K, cfg.width, cfg.height)
    alphas = eng.blend_weights(view, want_alphas=True)  # same view again: blend only
    tot_d, tot_a = float(d.double().sum()), float(alphas.double().sum())
    assert abs(tot_d - tot_a) <= 2e-5 * tot_a
    assert st["n_pairs"] > 5e7 and st["n_isect"] > 3e6 and st["n_visible"] > 7e5
    assert float((F - d[:, None]).abs().max()) <= 1e-4 * float(d.max())
    assert float((F.min(dim=1).values - F.max(dim=1).values).abs().max()) <= 1e-4 * float(d.max())
    assert bool((d >= 0).all()) and int((d > 0).sum()) < st["n_visible"]


def test_linearity_in_the_feature_map(c2, dev):
    cfg, eng, g, vms, K = c2
    D = 128
    gen = torch.Generator(device=dev).manual_seed(1)
    f1 = torch.randn(cfg.height, cfg.width, D, generator=gen, device=dev)
    f2 = torch.randn(cfg.height, cfg.width, D, generator=gen, device=dev)
    F1, d1, _ = _run(eng, cfg, g, vms[1], K, f1, dev)
    F2, d2, _ = _run(eng, cfg, g, vms[1], K, f2, dev)
    F3, d3, _ = _run(eng, cfg, g, vms[1], K, 0.5 * f1 - 2.0 * f2, dev)
    assert torch.equal(d1, d2) or float((d1 - d2).abs().max()) <= 1e-5 * float(d1.max())
    ref = 0.5 * F1 - 2.0 * F2
    scale = float(ref.norm(dim=1).max())
    assert float((F3 - ref).norm(dim=1).max()) <= 2e-5 * scale


def test_full_geometry_against_oracle(c2, dev, orc):
    """One C2 view (1M Gaussians, 1600x1060, D = 128: the fast scatter path) against the CPU oracle."""
    cfg, eng, g, vms, K = c2
    D = 128
    feats = syn.make_feature_map(cfg, 3, device=dev, dim=D)
    F, d, st = _run(eng, cfg, g, vms[0], K, feats, dev)
    h = [t.cpu().numpy() for t in g]
    Fr = np.zeros((cfg.n_gaussians, D), np.float32)
    dr = np.zeros(cfg.n_gaussians, np.float32)
    info = orc.backproject_view(*h, vms[0].numpy(), K.numpy(), cfg.width, cfg.height, feats.cpu().numpy(), Fr, dr)
    assert st["n_pairs"] == info["n_pairs"] and st["n_isect"] == info["n_isect"] and st["n_visible"] == info["n_vis"]
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4


def test_wide_and_narrow_scatter_kernels_agree_at_full_size(c2, dev):
    """C2 geometry, D = 256: the 256-channel kernel (half-tile slabs, carried partial sums, d through the headers'
    weight sums) against the 128-channel kernel on the same view; both sum the same bit-identical weights."""
    cfg, eng, g, vms, K = c2
    D = 256
    feats = syn.make_feature_map(cfg, 5, device=dev, dim=D)
    out = {}
    for name, narrow in (("narrow", True), ("wide", False)):
        eng.set_narrow_scatter(narrow)
        F, d, st = _run(eng, cfg, g, vms[1], K, feats, dev)
        out[name] = (F, d, st)
    eng.set_narrow_scatter(True)
    assert out["wide"][2]["n_pairs"] == out["narrow"][2]["n_pairs"]
    Fn, Fw = out["narrow"][0], out["wide"][0]
    scale = float(Fn.norm(dim=1).max())
    assert float((Fw - Fn).norm(dim=1).max()) <= 2e-5 * scale
    dn, dw = out["narrow"][1], out["wide"][1]
    assert float((dw - dn).abs().max()) <= 2e-5 * float(dn.max())
