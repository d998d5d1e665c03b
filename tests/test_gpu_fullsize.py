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


# ---- BASELINE.json configs at their own sizes, through the driver that ships ------------------------------------------
def _oracle_views(orc, cfg, g_host, vms, K, maps_host, D, rows=None, acc=np.float64):
    """Oracle accumulators over the given views; rows = int32[N] row map (subset) or None."""
    n_out = cfg.n_gaussians if rows is None else int(rows.max()) + 1
    Fr, dr = np.zeros((n_out, D), acc), np.zeros(n_out, acc)
    pairs = 0
    for v in range(len(maps_host)):
        info = orc.backproject_view(*g_host, vms[v].numpy(), K.numpy(), cfg.width, cfg.height, maps_host[v](), Fr, dr,
                                    row_of=rows)
        pairs += info["n_pairs"]
    return Fr, dr, pairs


def _spy_pipeline(monkeypatch):
    """Records which scatter kernel the ViewPipeline chose and that it really ran on two streams."""
    seen = {"choices": [], "fronts": 0}
    orig_choose, orig_front = gsbp_amd.ViewPipeline.choose_scatter_kernel, gsbp_amd.ViewPipeline.front

    def choose(self, a, b):
        r = orig_choose(self, a, b)
        seen["choices"].append(r)
        return r

    def front(self, *a, **k):
        seen["fronts"] += 1
        assert self.side != torch.cuda.current_stream(self.dev)
        return orig_front(self, *a, **k)

    monkeypatch.setattr(gsbp_amd.ViewPipeline, "choose_scatter_kernel", choose)
    monkeypatch.setattr(gsbp_amd.ViewPipeline, "front", front)
    return seen


def test_c2_d512_pipelined_wide_path_against_oracle(dev, orc, monkeypatch):
    """What bench.py times, checked at its own size: C2 (1M Gaussians, 1600x1060), D = 512 (two 256-channel chunks),
    three views through create_feature_field(pipeline=True): ViewPipeline on two streams, k_scatter_wide, the
    denominators added by the blend itself on the side stream (gwbp_blend_weights_d), raised front priority -- against the CPU oracle."""
    cfg = syn.CONFIGS["C2"]
    D, V = cfg.feat_dim, 3
    g_cpu = syn.activate(syn.make_scene(cfg))
    g = [t.to(dev) for t in g_cpu]
    vms, K = syn.make_cameras(cfg, n_views=V), syn.intrinsics(cfg)
    maps = [syn.make_feature_map(cfg, 20 + v, device=dev) for v in range(V)]
    seen = _spy_pipeline(monkeypatch)
    out, F, d, st = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height, lambda v: maps[v],
                                                  D, return_partials=True)
    assert st["overflow"] == 0 and seen["fronts"] == V and seen["choices"] == ["wide", "wide"]
    Fr, dr, pairs = _oracle_views(orc, cfg, [t.numpy() for t in g_cpu], vms, K,
                                  [lambda v=v: maps[v].cpu().numpy() for v in range(V)], D)
    assert st["n_pairs"] == pairs
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    assert np.abs(out.cpu().numpy() - orc.finalize(Fr, dr)).max() <= 1e-4
    # conservation: sum_g d[g] == sum over views and pixels of (1 - T)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
    tot_a = 0.0
    for v in range(V):
        view = eng.view(vms[v], K, cfg.width, cfg.height)
        eng.project(view, *g)
        eng.bin_sort(view)
        tot_a += float(eng.blend_weights(view, want_alphas=True).double().sum())
    assert abs(float(d.double().sum()) - tot_a) <= 2e-5 * tot_a


def test_c4_full_size_invariants_and_oracle_subset(dev, orc, monkeypatch):
    """BASELINE.json configs[3]: 5M Gaussians, D = 768 (F = 15.4 GB).  (i) one view of a channel-constant map through
    the fused call: conservation, F[:, c] == d, no overflow with the default capacities; (ii) wide vs narrow scatter
    kernel on one view; (iii) three views through the pipelined driver (which switches from the 256- to the
    128-channel kernel after two views: short records) against the oracle on a 64 k-Gaussian subset of the rows
    (SURVEY.md 8(d))."""
    cfg = syn.CONFIGS["C4"]
    N, D, V = cfg.n_gaussians, cfg.feat_dim, 3
    g_cpu = syn.activate(syn.make_scene(cfg))
    g = [t.to(dev) for t in g_cpu]
    vms, K = syn.make_cameras(cfg, n_views=V), syn.intrinsics(cfg)
    eng = gsbp_amd.Engine(N, cfg.width, cfg.height, device=dev, tight_binning=True)
    assert eng.ws_bytes < 40e9  # the default capacities (16 intersections per Gaussian) fit comfortably in 288 GB
    view = eng.view(vms[0], K, cfg.width, cfg.height)
    # (i)
    ones = torch.ones(cfg.height, cfg.width, D, device=dev)
    F = torch.zeros(N, D, device=dev)
    d = torch.zeros(N, device=dev)
    eng.backproject_view(view, *g, ones, F, d)
    st = eng.stats()
    assert st["overflow"] == 0 and st["n_visible"] > 3e6 and st["n_pairs"] > 5e7
    alphas = eng.blend_weights(view, want_alphas=True)
    tot_d, tot_a = float(d.double().sum()), float(alphas.double().sum())
    assert abs(tot_d - tot_a) <= 2e-5 * tot_a
    assert float((F - d[:, None]).abs().max()) <= 1e-4 * float(d.max())
    del ones, alphas
    # (ii)
    feats = syn.make_feature_map(cfg, 31, device=dev)
    eng.set_narrow_scatter(True)
    F.zero_(), d.zero_()
    eng.backproject_view(view, *g, feats, F, d)
    eng.set_narrow_scatter(False)
    Fw, dw = torch.zeros_like(F), torch.zeros_like(d)
    eng.backproject_view(view, *g, feats, Fw, dw)
    assert eng.stats()["overflow"] == 0
    scale = float(F.norm(dim=1).max())
    assert float((Fw - F).norm(dim=1).max()) <= 2e-5 * scale
    assert float((dw - d).abs().max()) <= 2e-5 * float(d.max())
    del F, Fw, d, dw, eng
    torch.cuda.empty_cache()
    # (iii)
    maps = [feats] + [syn.make_feature_map(cfg, 32 + v, device=dev) for v in range(1, V)]
    seen = _spy_pipeline(monkeypatch)
    out, F, d, st = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height, lambda v: maps[v],
                                                  D, return_partials=True)
    # the driver starts wide and re-decides from the counters of views 0 and 1 (C4: short records -> 128-channel kernel)
    expect = "wide" if st["n_pairs"] / st["n_headers"] >= gsbp_amd.ViewPipeline.WIDE_MIN_PAIRS_PER_RECORD else "narrow"
    assert st["overflow"] == 0 and seen["choices"] == ["wide", expect]
    sel = np.arange(0, N, N // 65536)[:65536]
    rows = np.full(N, -1, np.int32)
    rows[sel] = np.arange(sel.size, dtype=np.int32)
    Fr, dr, pairs = _oracle_views(orc, cfg, [t.numpy() for t in g_cpu], vms, K,
                                  [lambda v=v: maps[v].cpu().numpy() for v in range(V)], D, rows=rows)
    assert st["n_pairs"] == pairs
    sel_t = torch.from_numpy(sel).to(dev)
    assert int((dr > 0).sum()) > 20000  # the subset really sees weight
    assert rel_row_err(F[sel_t].cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d[sel_t].cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    assert np.abs(out[sel_t].cpu().numpy() - orc.finalize(Fr, dr)).max() <= 1e-4


@pytest.mark.parametrize("split", [True, False], ids=["producer_consumer", "one_wave_per_tile"])
def test_c5_full_size_encoder_path_against_oracle(dev, orc, split):
    """BASELINE.json configs[4] (backproject_compressed.py:127-165): 1M Gaussians, 512 -> 16 encoder, two full-size
    views through the pipelined driver; the oracle gets the map encoded on the CPU.  The fused kernel in both forms: encoder
    waves + blend waves of one persistent launch (round 6, GWBP_FLAG_SPLIT_ENCODER) and one wave per tile (round 5)."""
    cfg = syn.CONFIGS["C5"]
    V = 2
    g_cpu = syn.activate(syn.make_scene(cfg))
    g = [t.to(dev) for t in g_cpu]
    vms, K = syn.make_cameras(cfg, n_views=V), syn.intrinsics(cfg)
    enc = syn.make_encoder(cfg)
    maps = [syn.make_feature_map(cfg, 40 + v, device=dev) for v in range(V)]
    out, F, d, st = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height, lambda v: maps[v],
                                                  cfg.feat_dim, encoder=enc.to(dev), return_partials=True, encoder_split=split)
    assert st["overflow"] == 0 and tuple(F.shape) == (cfg.n_gaussians, cfg.encoder_dim)
    # the same job with the encoder fused into the scatter kernel's slab staging (gwbp_scatter_encoded)
    _, F2, d2, st2 = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height, lambda v: maps[v],
                                                   cfg.feat_dim, encoder=enc.to(dev), return_partials=True,
                                                   fuse_encoder=True)
    assert st2["overflow"] == 0 and st2["n_pairs"] == st["n_pairs"]
    assert float((F2 - F).norm(dim=1).max()) <= 2e-5 * float(F.norm(dim=1).max())
    assert float((d2 - d).abs().max()) <= 2e-5 * float(d.max())
    # ... and as separate blend (weight store) + small-D scatter kernels instead of the fused gwbp_blend_scatter
    _, F3, d3, st3 = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height, lambda v: maps[v],
                                                   cfg.feat_dim, encoder=enc.to(dev), return_partials=True,
                                                   fuse_small=False)
    assert st3["overflow"] == 0 and st3["n_pairs"] == st["n_pairs"] and st3["n_headers"] == st["n_headers"]
    assert float((F3 - F).norm(dim=1).max()) <= 2e-5 * float(F.norm(dim=1).max())
    assert float((d3 - d).abs().max()) <= 2e-5 * float(d.max())
    del F2, d2, F3, d3
    Fr, dr, pairs = _oracle_views(orc, cfg, [t.numpy() for t in g_cpu], vms, K,
                                  [lambda v=v: (maps[v].cpu() @ enc).numpy() for v in range(V)], cfg.encoder_dim)
    assert st["n_pairs"] == pairs
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    assert np.abs(out.cpu().numpy() - orc.finalize(Fr, dr)).max() <= 1e-4


@pytest.mark.parametrize("token_space", [True, False], ids=["token_space", "pixel_slabs"])
def test_dino_width_1024_nearest_64x64_mean_reduction(dev, orc, token_space):
    """create_feature_field_dino at its real width (backproject.py:201-210,242-289): D = 1024, a 64x64 patch-token map
    nearest-upsampled to the view, .mean() reductions; C2 geometry (1M Gaussians), two views, against the oracle fed the
    materialised F.interpolate(mode="nearest") map, on a 64 k-row subset.  Both product paths: TOKEN space (round 6, the default
    for this shape: per-record token-quadrant weight sums + one plain read-modify-write per F row, csrc/token.hip) and the
    pixel-slab kernels with index maps (gwbp_scatter_upsampled)."""
    cfg = syn.CONFIGS["C2"]
    N, D, V = cfg.n_gaussians, 1024, 2
    g_cpu = syn.activate(syn.make_scene(cfg))
    g = [t.to(dev) for t in g_cpu]
    vms, K = syn.make_cameras(cfg, n_views=V), syn.intrinsics(cfg)
    gen = torch.Generator().manual_seed(77)
    low = [torch.randn(64, 64, D, generator=gen) for _ in range(V)]
    low_dev = [t.to(dev) for t in low]
    out, F, d, st = gsbp_amd.create_feature_field(*g, vms.to(dev), K.to(dev), cfg.width, cfg.height,
                                                  lambda v: low_dev[v], D, reduction="mean", upsample="nearest",
                                                  return_partials=True, token_space=token_space)
    assert st["overflow"] == 0
    sel = np.arange(0, N, N // 65536)[:65536]
    rows = np.full(N, -1, np.int32)
    rows[sel] = np.arange(sel.size, dtype=np.int32)

    def up(v):  # the reference's own op (backproject.py:244-248)
        x = low[v].permute(2, 0, 1)[None]
        return torch.nn.functional.interpolate(x, size=(cfg.height, cfg.width), mode="nearest")[0].permute(1, 2, 0) \
            .contiguous().numpy()

    Fr, dr, pairs = _oracle_views(orc, cfg, [t.numpy() for t in g_cpu], vms, K, [lambda v=v: up(v) for v in range(V)],
                                  D, rows=rows)
    assert st["n_pairs"] == pairs
    Fr /= float(cfg.height * cfg.width * D)  # backproject.py:263
    dr /= float(cfg.height * cfg.width * 3)  # backproject.py:283
    sel_t = torch.from_numpy(sel).to(dev)
    assert rel_row_err(F[sel_t].cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d[sel_t].cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    assert np.abs(out[sel_t].cpu().numpy() - orc.finalize(Fr, dr)).max() <= 1e-4


def test_wide_forward_render_c2_band_vs_oracle(orc, c2, dev):
    """gwbp_render (k_render_rows) at C2 geometry with a 512-wide colour table -- the feature render of the reference's
    consumers (segment.py:209-220: rasterization(..., features[N,512], ...)) -- against the CPU oracle on a 64-row pixel
    band: out[p, :] = sum_g w_g(p) colors[g, :] with the ORACLE's own blend (its (Gaussian, pixel, weight) list of the
    view), accumulated in float64."""
    cfg, eng, g, vms, K = c2
    D = 512
    gen = torch.Generator().manual_seed(77)
    colors = torch.randn(cfg.n_gaussians, D, generator=gen)
    view = eng.view(vms[0], K, cfg.width, cfg.height)
    eng.project(view, *g)
    eng.bin_sort(view)
    eng.blend_weights(view)
    out = eng.render(view, colors.to(dev))
    assert eng.stats()["overflow"] == 0 and tuple(out.shape) == (cfg.height, cfg.width, D)
    y0, rows = 500, 64
    band = out[y0:y0 + rows].cpu().double().reshape(-1, D)
    h = [t.cpu().numpy() for t in g]
    proj = orc.project(h[0], h[1], h[2], vms[0].numpy(), K.numpy(), cfg.width, cfg.height)
    bins = orc.bin_sort(proj, cfg.width, cfg.height)
    gid, pix, w, _ = orc.blend_pairs(proj, bins, h[3], cfg.width, cfg.height)
    sel = (pix >= y0 * cfg.width) & (pix < (y0 + rows) * cfg.width)
    gid_b = torch.from_numpy(gid[sel].astype(np.int64))
    pix_b = torch.from_numpy((pix[sel] - y0 * cfg.width).astype(np.int64))
    w_b = torch.from_numpy(w[sel].astype(np.float64))
    assert len(gid_b) > 1e6
    ref = torch.zeros(rows * cfg.width, D, dtype=torch.float64)
    cd = colors.double()
    for s in range(0, len(gid_b), 1 << 18):  # 256 K pairs x 512 doubles = 1 GB of temporaries per slice
        e = s + (1 << 18)
        ref.index_add_(0, pix_b[s:e], w_b[s:e, None] * cd[gid_b[s:e]])
    scale = float(ref.abs().max())
    assert float((band - ref).abs().max()) <= 2e-6 * scale * 8  # ~50 terms per pixel, fp32 accumulators
    # pixels nobody covers render exactly zero
    empty = torch.ones(rows * cfg.width, dtype=torch.bool)
    empty[pix_b] = False
    assert float(band[empty].abs().max() if empty.any() else 0.0) == 0.0
