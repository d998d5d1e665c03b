"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bars: integer/index outputs and per-pair weights bit-exact; fp32 sums within 1e-4 relative (north_star),
measured per Gaussian row as in SURVEY.md 8(d).
"""
import os

import numpy as np
import pytest
import torch

from util import npy, rel_row_err, scene_np, sort_pairs, to_dev

import gsbp_amd
from gsbp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: "within 1e-4 relative fp32"


def _front(eng, sc, cfg, v, want=True):
    view = eng.view(sc["vms"][v], sc["K"], cfg.width, cfg.height)
    proj = eng.project(view, sc["means"], sc["quats"], sc["scales"], sc["opac"], want_outputs=want)
    bins = eng.bin_sort(view, want_outputs=want)
    return view, proj, bins


@pytest.mark.parametrize("name", ["T0", "T1", "C1"])
def test_projection_bit_exact(name, orc, dev):
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    for v in range(cfg.n_views):
        _, proj, _ = _front(eng, d, cfg, v)
        ref = orc.project(h["means"], h["quats"], h["scales"], h["vms"][v], h["K"], cfg.width, cfg.height)
        assert np.array_equal(proj["radii"].cpu().numpy(), ref["radii"])
        for k in ("means2d", "depths", "conics"):
            a, b = proj[k].cpu().numpy(), ref[k]
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{k} differs bitwise (view {v})"


@pytest.mark.parametrize("name,n_over", [("T0", None), ("T1", None), ("C1", None), ("T1", 5000), ("T1", 9000),
                                         ("C1", 12288), ("C1", 12289), ("C1", 30000)])
def test_bin_sort_exact(name, n_over, orc, dev):
    """n_over: other Gaussian counts on the same geometry -- up to 12288 the depth sort is ONE workgroup (k_sort_small with
    4, 8 or 12 items per thread), above it the multi-block radix passes."""
    cfg, sc = scene_np(name, **({"n_gaussians": n_over, "n_views": 1} if n_over else {}))
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    for v in range(cfg.n_views):
        _, _, bins = _front(eng, d, cfg, v)
        st = eng.stats()
        ref_p = orc.project(h["means"], h["quats"], h["scales"], h["vms"][v], h["K"], cfg.width, cfg.height)
        ref = orc.bin_sort(ref_p, cfg.width, cfg.height)
        n = ref["n_isect"]
        assert st["n_isect"] == n and st["overflow"] == 0
        assert st["n_visible"] == int((ref_p["radii"] > 0).sum())
        assert np.array_equal(bins["isect_ids"][:n].cpu().numpy(), ref["isect_ids"])
        assert np.array_equal(bins["flatten_ids"][:n].cpu().numpy(), ref["flatten_ids"])
        assert np.array_equal(bins["tile_offsets"].cpu().numpy(), ref["tile_offsets"])


@pytest.mark.parametrize("name", ["T0", "T1", "C1"])
def test_blend_weights_bit_exact(name, orc, dev):
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    v = 0
    view, _, _ = _front(eng, d, cfg, v, want=False)
    alphas = eng.blend_weights(view, want_alphas=True)
    gid, pix, w = eng.dump_pairs(view)
    ref_p = orc.project(h["means"], h["quats"], h["scales"], h["vms"][v], h["K"], cfg.width, cfg.height)
    ref_b = orc.bin_sort(ref_p, cfg.width, cfg.height)
    rg, rp, rw, ralpha = orc.blend_pairs(ref_p, ref_b, h["opac"], cfg.width, cfg.height, want_alphas=True)
    assert eng.stats()["n_pairs"] == len(rg)
    k1, w1 = sort_pairs(gid.cpu().numpy(), pix.cpu().numpy(), w.cpu().numpy())
    k2, w2 = sort_pairs(rg, rp, rw)
    assert np.array_equal(k1, k2), "set of contributing (gaussian, pixel) pairs differs"
    assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32)), "weights differ bitwise"
    assert np.array_equal(alphas.cpu().numpy().view(np.uint32), ralpha.view(np.uint32))


@pytest.mark.parametrize("name,D,wide", [("T0", 8, False), ("T0", 3, False), ("T1", 24, False), ("T1", 130, False),
                                         ("C1", 32, False), ("T1", 512, False), ("T1", 512, True), ("C1", 256, True)])
def test_scatter_parity(name, D, wide, orc, dev):
    """wide: the 256-channel kernel (an Engine starts with the 128-channel one; the drivers switch per job)."""
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    eng.set_narrow_scatter(not wide)
    F = torch.zeros(cfg.n_gaussians, D, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    pairs = 0
    for v in range(cfg.n_views):
        feats = syn.make_feature_map(cfg, v, dim=D)
        view = eng.view(d["vms"][v], d["K"], cfg.width, cfg.height)
        eng.backproject_view(view, d["means"], d["quats"], d["scales"], d["opac"], feats.to(dev), F, dd)
        st = eng.stats()
        assert st["overflow"] == 0
        info = orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][v], h["K"], cfg.width,
                                    cfg.height, feats.numpy(), Fr, dr)
        assert st["n_pairs"] == info["n_pairs"] and st["n_isect"] == info["n_isect"]
        pairs += info["n_pairs"]
    assert pairs > 0
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL
    assert rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
    out = eng.finalize(F, dd).cpu().numpy()
    ref = orc.finalize(Fr, dr)
    assert np.abs(out - ref).max() <= TOL
    assert np.array_equal(out[dr == 0], np.zeros_like(out[dr == 0]))  # NaN -> 0 rows (backproject.py:169)


@pytest.mark.parametrize("name,D", [("T0", 16), ("T0", 3), ("T1", 16), ("T1", 5), ("T1", 8), ("C1", 16), ("C1", 1),
                                    ("T1", 32), ("T1", 17), ("C1", 32), ("C1", 24)])
def test_blend_scatter_fused_parity(name, D, orc, dev):
    """gwbp_blend_scatter: blend and scatter in one kernel, no weight store -- same F, d, alphas, pair count.  These images
    have few tiles: the wave-per-quarter-tile form (D <= 32); the wave-per-tile form (D <= 16) runs at C5 size in
    test_gpu_fullsize.py and in test_blend_scatter_wave_per_tile_form below."""
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    F = torch.zeros(cfg.n_gaussians, D, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    for v in range(cfg.n_views):
        feats = syn.make_feature_map(cfg, v, dim=D)
        if D < 4:  # a row of one to three signed channels can cancel to nearly nothing: the relative error of such a row
            feats = feats.abs()  # measures the summation order, not the kernel (tools/fuzz_parity.py does the same)
        fd = feats.to(dev)
        if v % 2 == 1:  # a pixel stride that is not a multiple of 4 floats: the scalar-load path
            wide = torch.zeros(cfg.height, cfg.width, D + 1, device=dev)
            wide[..., :D] = fd
            fd = wide[..., :D]
        view = eng.view(d["vms"][v], d["K"], cfg.width, cfg.height)
        eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
        eng.bin_sort(view)
        alphas = eng.blend_scatter(view, fd, F, dd, want_alphas=True)
        st = eng.stats()
        assert st["overflow"] == 0
        ref_p = orc.project(h["means"], h["quats"], h["scales"], h["vms"][v], h["K"], cfg.width, cfg.height)
        ref_b = orc.bin_sort(ref_p, cfg.width, cfg.height)
        _, _, _, ralpha = orc.blend_pairs(ref_p, ref_b, h["opac"], cfg.width, cfg.height, want_alphas=True)
        info = orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][v], h["K"], cfg.width,
                                    cfg.height, feats.numpy(), Fr, dr)
        assert st["n_pairs"] == info["n_pairs"] and st["n_isect"] == info["n_isect"]
        assert np.array_equal(alphas.cpu().numpy().view(np.uint32), ralpha.view(np.uint32))
        # the store is empty: scattering the view again adds nothing
        F2, d2 = torch.zeros_like(F), torch.zeros_like(dd)
        eng.scatter(view, fd, F2, d2)
        assert not F2.any() and not d2.any()
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL
    assert rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL


def test_blend_scatter_wave_per_tile_form(orc, dev):
    """An image of more than 4096 tiles (1040 x 1040 = 65 x 65) takes the one-wave-per-tile fused kernel (D <= 16)."""
    cfg, sc = scene_np("T1", width=1040, height=1040, n_views=1)
    d, h = to_dev(sc, dev), npy(sc)
    D = 16
    assert gsbp_amd.Engine.fused_max_dim(cfg.width, cfg.height) == 16
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, isect_cap=1 << 23)
    feats = syn.make_feature_map(cfg, 0, dim=D)
    F, dd = torch.zeros(cfg.n_gaussians, D, device=dev), torch.zeros(cfg.n_gaussians, device=dev)
    view = eng.view(d["vms"][0], d["K"], cfg.width, cfg.height)
    eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
    eng.bin_sort(view)
    eng.blend_scatter(view, feats.to(dev), F, dd)
    st = eng.stats()
    Fr, dr = np.zeros((cfg.n_gaussians, D), np.float64), np.zeros(cfg.n_gaussians, np.float64)
    info = orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height,
                                feats.numpy(), Fr, dr)
    assert st["overflow"] == 0 and st["n_pairs"] == info["n_pairs"] > 0
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL
    assert rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
    with pytest.raises(gsbp_amd.GwbpError):  # 32 channels only on small images
        eng.blend_scatter(view, torch.zeros(cfg.height, cfg.width, 32, device=dev), torch.zeros(cfg.n_gaussians, 32, device=dev), dd)


def test_scatter_strided_feature_map(orc, dev):
    """backproject.py:113 hands a permuted [D,H,W] view as feats; strides travel through the C ABI."""
    cfg, sc = scene_np("T0")
    d, h = to_dev(sc, dev), npy(sc)
    D = 8
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    feats = syn.make_feature_map(cfg, 0, dim=D)
    planar = feats.permute(2, 0, 1).contiguous().to(dev)  # [D,H,W]
    F1 = torch.zeros(cfg.n_gaussians, D, device=dev)
    F2 = torch.zeros_like(F1)
    d1 = torch.zeros(cfg.n_gaussians, device=dev)
    d2 = torch.zeros_like(d1)
    view = eng.view(d["vms"][0], d["K"], cfg.width, cfg.height)
    eng.backproject_view(view, d["means"], d["quats"], d["scales"], d["opac"], planar.permute(1, 2, 0), F1, d1)
    eng.backproject_view(view, d["means"], d["quats"], d["scales"], d["opac"], feats.to(dev), F2, d2)
    assert rel_row_err(F1.cpu().numpy(), F2.cpu().numpy()) <= 1e-5


def test_render_forward_parity(orc, dev):
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    for D in (3, 20, 128, 132, 256, 260, 512, 708):  # (D >= 128, D % 4 == 0: k_render_rows4 -- 512 channels per wave for whole blocks of 512,
        # 256 per wave for the rest, a partial last block)
        colors = torch.rand(cfg.n_gaussians, D, generator=torch.Generator().manual_seed(5))
        view, _, _ = _front(eng, d, cfg, 1, want=False)
        eng.blend_weights(view)
        out = eng.render(view, colors.to(dev)).cpu().numpy()
        ref_p = orc.project(h["means"], h["quats"], h["scales"], h["vms"][1], h["K"], cfg.width, cfg.height)
        ref_b = orc.bin_sort(ref_p, cfg.width, cfg.height)
        ref, _ = orc.render(ref_p, ref_b, h["opac"], colors.numpy(), cfg.width, cfg.height)
        assert np.abs(out - ref).max() <= 1e-5


@pytest.mark.parametrize("D", [None, 256], ids=["D8", "D256_wide_scatter_in_backward"])
def test_reference_loop_through_dropin_shim(orc, dev, D):
    """The literal loop of create_feature_field_lseg (backproject.py:62-72,115-151,166-169) with
    `rasterization` swapped for the drop-in -- zeros colours, .backward(), colors.grad harvest."""
    from gsbp_amd import rasterization
    cfg, sc = scene_np("T0")
    d, h = to_dev(sc, dev), npy(sc)
    N, D = cfg.n_gaussians, (D or cfg.feat_dim)
    gaussian_features = torch.zeros(N, D, device=dev)
    gaussian_denoms = torch.ones(N, device=dev) * 1e-12
    colors_feats = torch.zeros(N, D, device=dev, requires_grad=True)
    colors_feats_0 = torch.zeros(N, 3, device=dev, requires_grad=True)
    feats_all = [syn.make_feature_map(cfg, v, dim=D) for v in range(cfg.n_views)]
    from gsbp_amd.rasterization import get_engine
    gen0 = get_engine(dev, N, cfg.width, cfg.height).generation  # (the engine outlives the test: count from here)
    for v in range(cfg.n_views):
        feats = feats_all[v].to(dev)
        out, _, meta = rasterization(d["means"], d["quats"], d["scales"], d["opac"], colors_feats, d["vms"][v][None],
                                     d["K"][None], width=cfg.width, height=cfg.height)
        assert float(out.abs().max()) == 0.0  # zero colours render to zero
        target = (out[0] * feats).sum()
        target.backward()
        colors_feats_copy = colors_feats.grad.clone()
        colors_feats.grad.zero_()
        out0, _, _ = rasterization(d["means"], d["quats"], d["scales"], d["opac"], colors_feats_0, d["vms"][v][None],
                                   d["K"][None], width=cfg.width, height=cfg.height)
        out0[0].sum().backward()
        gaussian_features += colors_feats_copy
        gaussian_denoms += colors_feats_0.grad[:, 0]
        # demo_affordance_transfer.py:383-386: channel-constant v_render => channel-equal gradients
        g0 = colors_feats_0.grad
        assert torch.allclose(g0.min(dim=1).values, g0.max(dim=1).values, rtol=1e-5, atol=1e-7)
        colors_feats_0.grad.zero_()
    gaussian_features = gaussian_features / gaussian_denoms[..., None]
    gaussian_features = gaussian_features / gaussian_features.norm(dim=-1, keepdim=True)
    gaussian_features[torch.isnan(gaussian_features)] = 0
    ref, _, _, _ = orc.backproject_oracle(h["means"], h["quats"], h["scales"], h["opac"], h["vms"], h["K"], cfg.width,
                                          cfg.height, lambda v: feats_all[v].numpy(), D)
    assert np.abs(gaussian_features.cpu().numpy() - ref).max() <= TOL
    assert meta["means2d"].shape[1] == 2 and meta["gaussian_ids"].dtype == torch.int64
    # the second rasterization() of a view (zeros [N,3]) found the first one's projection / lists / weight store in the
    # workspace: one front pass per view, not two
    assert get_engine(dev, N, cfg.width, cfg.height).generation == gen0 + cfg.n_views
    # ... and an in-place change of the Gaussians is noticed
    d["means"].add_(0.0)
    rasterization(d["means"], d["quats"], d["scales"], d["opac"], colors_feats_0, d["vms"][0][None], d["K"][None],
                  width=cfg.width, height=cfg.height)
    assert get_engine(dev, N, cfg.width, cfg.height).generation == gen0 + cfg.n_views + 1


def test_fused_driver_matches_oracle_mean_and_encoder(orc, dev):
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    feats_all = [syn.make_feature_map(cfg, v) for v in range(cfg.n_views)]
    out = gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], d["vms"], d["K"], cfg.width,
                                        cfg.height, lambda v: feats_all[v].to(dev), cfg.feat_dim, reduction="mean")
    ref, _, _, _ = orc.backproject_oracle(h["means"], h["quats"], h["scales"], h["opac"], h["vms"], h["K"], cfg.width,
                                          cfg.height, lambda v: feats_all[v].numpy(), cfg.feat_dim, reduction="mean")
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL
    # encoder (backproject_compressed.py:127): through the hand-written GEMM, whose domain wants K % 16 == 0 ...
    K_in = 32
    maps32 = [syn.make_feature_map(cfg, v, dim=K_in) for v in range(cfg.n_views)]
    enc = torch.randn(K_in, 16, generator=torch.Generator().manual_seed(7)) / K_in ** 0.5
    out = gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], d["vms"], d["K"], cfg.width,
                                        cfg.height, lambda v: maps32[v].to(dev), K_in, encoder=enc.to(dev))
    ref, _, _, _ = orc.backproject_oracle(h["means"], h["quats"], h["scales"], h["opac"], h["vms"], h["K"], cfg.width,
                                          cfg.height, lambda v: (maps32[v] @ enc).numpy(), 16)
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL
    # ... and a shape outside it (K = 24) is REFUSED by every schedule of the driver: no library GEMM behind the caller's back
    enc24 = torch.randn(cfg.feat_dim, 16, generator=torch.Generator().manual_seed(7)).to(dev)
    for kw in (dict(), dict(pipeline=False), dict(fuse_encoder=True)):
        with pytest.raises(gsbp_amd.GwbpError, match="encode_map"):
            gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], d["vms"], d["K"], cfg.width,
                                          cfg.height, lambda v: feats_all[v].to(dev), cfg.feat_dim, encoder=enc24, **kw)


def test_empty_and_degenerate_inputs(dev):
    """No Gaussian visible (all behind the camera) -> F, d untouched, out all zeros; tiny capacity -> overflow flag."""
    cfg, sc = scene_np("T0")
    d = to_dev(sc, dev)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    vm = d["vms"][0].clone()
    vm[2, 3] = -100.0  # push everything behind the near plane
    F = torch.zeros(cfg.n_gaussians, 8, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    feats = syn.make_feature_map(cfg, 0, dim=8).to(dev)
    eng.backproject_view(eng.view(vm, d["K"], cfg.width, cfg.height), d["means"], d["quats"], d["scales"], d["opac"],
                         feats, F, dd)
    st = eng.stats()
    assert st["n_visible"] == 0 and st["n_isect"] == 0 and st["n_pairs"] == 0
    assert float(F.abs().max()) == 0.0 and float(eng.finalize(F, dd).abs().max()) == 0.0
    small = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, isect_cap=64)
    small.backproject_view(small.view(d["vms"][0], d["K"], cfg.width, cfg.height), d["means"], d["quats"],
                           d["scales"], d["opac"], feats, F, dd)
    assert small.stats()["overflow"] & 1


def test_rerun_determinism_bound(dev):
    """Atomic order is the only nondeterminism: two runs agree to fp32 reorder noise."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    feats = syn.make_feature_map(cfg, 0).to(dev)
    res = []
    for _ in range(2):
        F = torch.zeros(cfg.n_gaussians, cfg.feat_dim, device=dev)
        dd = torch.zeros(cfg.n_gaussians, device=dev)
        eng.backproject_view(eng.view(d["vms"][0], d["K"], cfg.width, cfg.height), d["means"], d["quats"],
                             d["scales"], d["opac"], feats, F, dd)
        res.append(F.cpu().numpy())
    assert rel_row_err(res[0], res[1]) <= 1e-5


def test_hip_path_reproduces_golden_vectors(dev):
    """tests/golden/g0.npz (committed inputs + expected outputs): projection, sort keys, per-pair weights and the
    alpha map bit for bit; F, d, out within the north_star tolerance."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g0.npz")))
    W0, H0, N, D = 64, 48, 256, 8
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("means", "quats", "scales", "opac")}
    eng = gsbp_amd.Engine(N, W0, H0, device=dev)
    view = eng.view(torch.from_numpy(g["vms"][0]), torch.from_numpy(g["K"]), W0, H0)
    proj = eng.project(view, t["means"], t["quats"], t["scales"], t["opac"], want_outputs=True)
    bins = eng.bin_sort(view, want_outputs=True)
    alphas = eng.blend_weights(view, want_alphas=True)
    for k in ("radii", "means2d", "conics", "depths"):
        assert np.array_equal(proj[k].cpu().numpy().view(np.uint32), g["v0_" + k].view(np.uint32)), k
    n = int(g["n_isect"][0])
    assert np.array_equal(bins["isect_ids"][:n].cpu().numpy(), g["v0_isect_ids"])
    assert np.array_equal(bins["flatten_ids"][:n].cpu().numpy(), g["v0_flatten_ids"])
    assert np.array_equal(bins["tile_offsets"].cpu().numpy(), g["v0_tile_offsets"])
    gid, pix, w = eng.dump_pairs(view)
    k1, w1 = sort_pairs(gid.cpu().numpy(), pix.cpu().numpy(), w.cpu().numpy())
    k2, w2 = sort_pairs(g["v0_pair_gid"], g["v0_pair_pix"], g["v0_pair_w"])
    assert np.array_equal(k1, k2) and np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    assert np.array_equal(alphas.cpu().numpy().view(np.uint32), g["v0_alphas"].view(np.uint32))
    out, F, d, st = gsbp_amd.create_feature_field(t["means"], t["quats"], t["scales"], t["opac"],
                                                   torch.from_numpy(g["vms"]), torch.from_numpy(g["K"]), W0, H0,
                                                   lambda v: torch.from_numpy(g["feats"][v]).to(dev), D, engine=eng,
                                                   return_partials=True)
    assert st["n_pairs"] == int(g["n_pairs"].sum())
    assert rel_row_err(F.cpu().numpy(), g["F"]) <= TOL and rel_row_err(d.cpu().numpy()[:, None], g["d"][:, None]) <= TOL
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL


def test_dropin_front_cache_is_not_fooled_by_a_recycled_address(dev):
    """rasterization() skips projection / sort / blend when it is called again for the same view with the SAME Gaussian
    tensors (the reference rasterises every view twice, backproject.py:115-143).  "Same" must mean the same tensor objects:
    a temporary such as `opac.clone()` with edited values can be handed the freed address (and version) of the previous
    call's temporary by the caching allocator -- a viewer or segment loop from a fixed camera does exactly that -- and must
    then NOT be served the stale weight store."""
    from gsbp_amd import rasterization
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    rgb = torch.rand(cfg.n_gaussians, 3, generator=torch.Generator().manual_seed(2)).to(dev)

    def render(opac):
        with torch.no_grad():
            out, alpha, _ = rasterization(d["means"], d["quats"], d["scales"], opac, rgb, d["vms"][0][None], d["K"][None],
                                          width=cfg.width, height=cfg.height, want_meta=False)
        return out.clone(), alpha.clone()

    o1 = d["opac"].clone()
    ptr1 = o1.data_ptr()
    out1, a1 = render(o1)
    del o1
    o2 = d["opac"].clone()   # usually lands on o1's freed block: same data_ptr, same _version
    o2[: cfg.n_gaussians // 2] = 0.0  # (in place: _version differs from a fresh clone's only by this one write)
    o3 = d["opac"].clone()
    recycled = o2.data_ptr() == ptr1
    out2, a2 = render(o2)
    # ground truth for the edited opacities through a tensor the cache has never seen
    o2b = o2.clone()
    out2b, a2b = render(o2b)
    assert torch.equal(out2, out2b) and torch.equal(a2, a2b)
    assert not torch.equal(a1, a2)  # half the Gaussians are gone: the alpha map must have changed
    # and the legitimate reuse still works: the same tensor object again gives the identical result
    out3, a3 = render(o2b)
    assert torch.equal(out3, out2b)
    assert recycled or o3 is not None  # (informational: the allocator normally recycles the address)


def test_sh_rgb_render_through_shim(orc, dev):
    """backproject.py:89-100: rasterization(..., colors_all [N,16,3], sh_degree=3) under no_grad; plus RGB+D
    (click_and_segment.py:251) and backgrounds.  SH colours and the pixel-parallel render vs the oracle."""
    from gsbp_amd import rasterization
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    N = cfg.n_gaussians
    g = torch.Generator().manual_seed(11)
    sh = torch.randn(N, 16, 3, generator=g) * 0.3
    v = 0
    vm = h["vms"][v]
    campos = -(vm[:3, :3].T @ vm[:3, 3])
    with torch.no_grad():
        out, alpha, meta = rasterization(d["means"], d["quats"], d["scales"], d["opac"], sh.to(dev), d["vms"][v][None],
                                         d["K"][None], width=cfg.width, height=cfg.height, sh_degree=3)
    cols_ref = orc.sh_colors(3, h["means"], sh.numpy(), campos)
    eng = gsbp_amd.Engine(N, cfg.width, cfg.height, device=dev)
    cols = eng.sh_colors(3, d["means"], sh.to(dev), campos.tolist()).cpu().numpy()
    assert np.abs(cols - cols_ref).max() < 2e-6
    ref_p = orc.project(h["means"], h["quats"], h["scales"], vm, h["K"], cfg.width, cfg.height)
    ref_b = orc.bin_sort(ref_p, cfg.width, cfg.height)
    ref, ralpha = orc.render(ref_p, ref_b, h["opac"], cols_ref, cfg.width, cfg.height)
    assert out.shape == (1, cfg.height, cfg.width, 3) and alpha.shape == (1, cfg.height, cfg.width, 1)
    assert np.abs(out[0].cpu().numpy() - ref).max() < 1e-5
    assert np.array_equal(alpha[0, :, :, 0].cpu().numpy().view(np.uint32), ralpha.view(np.uint32))
    # RGB+D with a background colour
    bg = torch.tensor([[0.2, 0.4, 0.6, 0.0]], device=dev)
    rgb = torch.rand(N, 3, generator=g)
    with torch.no_grad():
        outd, alphad, _ = rasterization(d["means"], d["quats"], d["scales"], d["opac"], rgb.to(dev), d["vms"][v][None],
                                        d["K"][None], cfg.width, cfg.height, render_mode="RGB+D", backgrounds=bg,
                                        want_meta=False)
    z = (h["means"] @ vm[:3, :3].T + vm[:3, 3])[:, 2:3]
    refd, _ = orc.render(ref_p, ref_b, h["opac"], np.concatenate([rgb.numpy(), z], 1).astype(np.float32), cfg.width,
                         cfg.height)
    refd = refd + (1.0 - ralpha[..., None]) * bg.cpu().numpy()[0]
    assert np.abs(outd[0].cpu().numpy() - refd).max() < 2e-5


@pytest.mark.parametrize("cfg_name", ["T1", "T0"])
def test_pixel_render_up_to_32_channels(orc, dev, cfg_name):
    """gwbp_render_pixels for 5..32 channels (round 5: segment_compressed.py:154-165 renders the 16-d compressed field per
    frame; no weight store needed): colours vs the oracle's render, the alpha map bit for bit, a non-finite colour only where
    its Gaussian has weight, the wide kernel's output for the same table, and the drop-in's route (D <= 32 -> this kernel)."""
    from gsbp_amd import rasterization
    cfg, sc = scene_np(cfg_name)
    d, h = to_dev(sc, dev), npy(sc)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    eng = gsbp_amd.Engine(N, W, H, device=dev)
    view, _, _ = _front(eng, d, cfg, 1, want=False)
    ref_p = orc.project(h["means"], h["quats"], h["scales"], h["vms"][1], h["K"], W, H)
    ref_b = orc.bin_sort(ref_p, W, H)
    for D in (1, 4, 5, 16, 20, 32):
        cols = torch.randn(N, D, generator=torch.Generator().manual_seed(D))
        out, alpha = eng.render_pixels(view, cols.to(dev))
        ref, ralpha = orc.render(ref_p, ref_b, h["opac"], cols.numpy(), W, H)
        assert out.shape == (H, W, D)
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5
        assert np.array_equal(alpha.cpu().numpy().view(np.uint32), ralpha.view(np.uint32))
    # the weight-store render of the same table: same order of additions, same fused multiply-adds
    cols = torch.randn(N, 16, generator=torch.Generator().manual_seed(99))
    out_px, _ = eng.render_pixels(view, cols.to(dev))
    eng.blend_weights(view)
    assert torch.equal(out_px, eng.render(view, cols.to(dev)))
    # a NaN colour reaches exactly the pixels its Gaussian has a weight at
    gid, pix, w, _ = orc.blend_pairs(ref_p, ref_b, h["opac"], W, H)
    g_bad = int(np.bincount(gid, minlength=N).argmax())
    cols_bad = cols.clone()
    cols_bad[g_bad, 3] = float("nan")
    out_bad, _ = eng.render_pixels(view, cols_bad.to(dev))
    hit = np.zeros(H * W, bool)
    hit[pix[gid == g_bad]] = True
    nan_map = torch.isnan(out_bad[..., 3]).cpu().numpy().reshape(-1)
    assert hit.any() and np.array_equal(nan_map, hit)
    assert not torch.isnan(out_bad[..., :3]).any() and not torch.isnan(out_bad[..., 4:]).any()
    with pytest.raises(gsbp_amd.GwbpError):
        eng.render_pixels(view, torch.zeros(N, 33, device=dev))
    with torch.no_grad():
        o, a, _ = rasterization(d["means"], d["quats"], d["scales"], d["opac"], cols.to(dev), d["vms"][1][None], d["K"][None],
                                width=W, height=H, want_meta=False)
    assert torch.equal(o[0], out_px) and o.shape == (1, H, W, 16)


def test_prune_mask_equals_reference_rule(orc, dev):
    """utils.prune_by_gradients (utils.py:222-271): keep Gaussians with accumulated |colour grad| > 0 == d > 0."""
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    feats_all = [syn.make_feature_map(cfg, v, dim=4) for v in range(cfg.n_views)]
    _, _, dd, _ = gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], d["vms"], d["K"],
                                                cfg.width, cfg.height, lambda v: feats_all[v].to(dev), 4,
                                                return_partials=True)
    _, _, dr, _ = orc.backproject_oracle(h["means"], h["quats"], h["scales"], h["opac"], h["vms"], h["K"], cfg.width,
                                         cfg.height, lambda v: feats_all[v].numpy(), 4)
    mask = gsbp_amd.prune_mask(dd).cpu().numpy()
    assert np.array_equal(mask, dr > 0) and 0 < mask.sum() < cfg.n_gaussians


def test_edge_cases_zero_gaussians_tiny_image_and_huge_splats(orc, dev):
    """N = 0; an image smaller than one tile; Gaussians covering the whole image (records with > 128 entries per
    tile exercise the rare multi-vector path); non-multiple-of-16 sizes."""
    # N = 0
    eng0 = gsbp_amd.Engine(0, 40, 24, device=dev)
    z3, z4, z1 = torch.zeros(0, 3, device=dev), torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev)
    F0, d0 = torch.zeros(0, 8, device=dev), torch.zeros(0, device=dev)
    view = eng0.view(torch.eye(4), torch.tensor([[50.0, 0, 20], [0, 50.0, 12], [0, 0, 1]]), 40, 24)
    eng0.backproject_view(view, z3, z4, z3, z1, torch.rand(24, 40, 8, device=dev), F0, d0)
    assert eng0.stats()["n_pairs"] == 0
    # 13 x 9 image (one partial tile), 3 huge + 20 small Gaussians, D = 5 (odd channel count)
    W, H, D = 13, 9, 5
    g = torch.Generator().manual_seed(3)
    n = 23
    means = torch.cat([torch.tensor([[0.0, 0.0, 2.0]] * 3), torch.rand(20, 3, generator=g) * torch.tensor([1.0, 0.6, 1.0])
                       + torch.tensor([-0.5, -0.3, 1.5])])
    scales = torch.cat([torch.full((3, 3), 2.0), torch.full((20, 3), 0.05)])
    quats = torch.randn(n, 4, generator=g)
    opac = torch.cat([torch.tensor([0.02, 0.3, 0.9]), torch.rand(20, generator=g)])
    K = torch.tensor([[15.0, 0, W / 2], [0, 15.0, H / 2], [0, 0, 1]])
    vm = torch.eye(4)
    feats = torch.randn(H, W, D, generator=g)
    eng = gsbp_amd.Engine(n, W, H, device=dev)
    F = torch.zeros(n, D, device=dev)
    dd = torch.zeros(n, device=dev)
    eng.backproject_view(eng.view(vm, K, W, H), means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev),
                         feats.to(dev), F, dd)
    Fr, dr = np.zeros((n, D)), np.zeros(n)
    info = orc.backproject_view(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(), vm.numpy(), K.numpy(), W, H,
                                feats.numpy(), Fr, dr)
    assert eng.stats()["n_pairs"] == info["n_pairs"] > 0
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL and rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
    # whole-screen splats at D = 128 through the fast path: every tile record has up to 256 entries
    W, H, D = 96, 64, 128
    K = torch.tensor([[80.0, 0, W / 2], [0, 80.0, H / 2], [0, 0, 1]])
    n = 6
    means = torch.tensor([[0.0, 0.0, 2.0 + 0.1 * i] for i in range(n)])
    scales = torch.full((n, 3), 3.0)
    quats = torch.randn(n, 4, generator=g)
    opac = torch.full((n,), 0.05)
    feats = torch.randn(H, W, D, generator=g)
    eng = gsbp_amd.Engine(n, W, H, device=dev)
    F = torch.zeros(n, D, device=dev)
    dd = torch.zeros(n, device=dev)
    eng.backproject_view(eng.view(vm, K, W, H), means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev),
                         feats.to(dev), F, dd)
    Fr, dr = np.zeros((n, D)), np.zeros(n)
    info = orc.backproject_view(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(), vm.numpy(), K.numpy(), W, H,
                                feats.numpy(), Fr, dr)
    assert eng.stats()["n_pairs"] == info["n_pairs"] == n * W * H  # every Gaussian reaches every pixel
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL and rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL


def test_pipelined_driver_equals_serial_driver(dev):
    cfg, sc = scene_np("T1", n_views=5)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=5).to(dev)
    feats_all = [syn.make_feature_map(cfg, v).to(dev) for v in range(5)]
    res = []
    for pipeline in (True, False):
        out, F, dd, st = gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"],
                                                        cfg.width, cfg.height, lambda v: feats_all[v], cfg.feat_dim,
                                                        pipeline=pipeline, return_partials=True)
        res.append((F.cpu().numpy(), dd.cpu().numpy(), st["n_pairs"]))
    assert res[0][2] == res[1][2]
    assert rel_row_err(res[0][0], res[1][0]) <= 1e-5 and np.abs(res[0][1] - res[1][1]).max() <= 1e-4 * res[1][1].max()


def test_driver_degrades_to_one_stream_when_the_queue_request_came_late(dev, monkeypatch):
    """a caller that touched CUDA before `import gsbp_amd` (a notebook, another library) got a hard GwbpError from
    create_feature_field(pipeline=True) for a concern that only affects speed.  The driver now warns and runs the views on one
    stream (same F and d); constructing a ViewPipeline explicitly still raises."""
    from gsbp_amd import _lib
    cfg, sc = scene_np("T1", n_views=3)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=3).to(dev)
    feats_all = [syn.make_feature_map(cfg, v).to(dev) for v in range(3)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: feats_all[v],
            cfg.feat_dim)
    ref = gsbp_amd.create_feature_field(*args, pipeline=False)
    monkeypatch.setattr(_lib, "_QUEUES_LATE", True)
    with pytest.warns(RuntimeWarning, match="ONE stream"):
        out = gsbp_amd.create_feature_field(*args)
    assert rel_row_err(out.cpu().numpy(), ref.cpu().numpy()) <= 1e-5
    with pytest.raises(gsbp_amd.GwbpError, match="GPU_MAX_HW_QUEUES"):
        gsbp_amd.ViewPipeline(cfg.n_gaussians, cfg.width, cfg.height, dev)


def test_driver_rebuilds_the_field_when_the_ring_kernel_reports_a_stall(dev, monkeypatch):
    """The producer / consumer form of the encoder-fused kernel bounds its LDS-ring waits and reports a wave that gave up through
    gwbp_stats.overflow bit 4 (never observed; the bound exists so that a scheduling accident cannot hang the device).  The driver
    used to raise; it now warns, zeroes the accumulators and builds the field again with the one-wave-per-tile form.  The stall
    is injected into the first pipeline's counters; the result must equal a run that never used the ring kernel."""
    from gsbp_amd import backproject as bp
    cfg, sc = scene_np("T1", n_views=4)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=4).to(dev)
    K_in, n_out = 64, 16
    g = torch.Generator().manual_seed(15)
    wide = [torch.randn(cfg.height, cfg.width, K_in, generator=g).to(dev) for _ in range(4)]
    enc = (torch.randn(K_in, n_out, generator=g) / K_in ** 0.5).to(dev)
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: wide[v], K_in)
    ref = gsbp_amd.create_feature_field(*args, encoder=enc, encoder_split=False, return_partials=True)
    real_stats, seen = bp.ViewPipeline.stats, []

    def stats(self):
        st = real_stats(self)
        seen.append(self.split_encoder)
        if self.split_encoder:
            st = dict(st, overflow=st["overflow"] | 16)
        return st

    monkeypatch.setattr(bp.ViewPipeline, "stats", stats)
    with pytest.warns(RuntimeWarning, match="encoder_split=False"):
        got = gsbp_amd.create_feature_field(*args, encoder=enc, encoder_split=True, return_partials=True)
    assert True in seen and seen[-1] is False  # the ring kernel ran, was reported stalled, the rerun did without it
    assert got[3]["overflow"] == 0
    assert rel_row_err(got[1].cpu().numpy(), ref[1].cpu().numpy()) <= 1e-5
    assert rel_row_err(got[2].cpu().numpy()[:, None], ref[2].cpu().numpy()[:, None]) <= 1e-5


def _capture_cases():
    from util import capture_tool
    return capture_tool().CASES


@pytest.mark.parametrize("fname,cfgname,dim,enc_dim", _capture_cases(), ids=[c[0] for c in _capture_cases()])
def test_hip_against_gsplat_capture(dev, fname, cfgname, dim, enc_dim):
    """The HIP path against a capture of REAL gsplat 1.4.0 output (tools/capture_gsplat_fixture.py), when one has been
    committed; skipped otherwise (parity unpinned).  One case per kernel family (general / fused small-D / encoder + fused /
    128-channel / 256-channel scatter).  Same bar as tests/test_oracle.py::test_oracle_against_gsplat_capture:
    99 % of the rows within the north_star 1e-4, at most 0.2 % threshold rows beyond it."""
    from util import capture_report, capture_tool
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fname)
    if not os.path.exists(path):
        pytest.skip(f"{fname} not captured yet (needs CUDA + gsplat==1.4.0)")
    cap = dict(np.load(path))
    if cfgname is None:
        g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g0.npz")))
        W, H = 64, 48
    else:
        cfg = syn.CONFIGS[cfgname]
        g = capture_tool().case_inputs(cfgname, dim, enc_dim)
        W, H = cfg.width, cfg.height
    t = {k: torch.from_numpy(np.asarray(g[k])).to(dev) for k in ("means", "quats", "scales", "opac", "K", "vms")}
    feats = [torch.from_numpy(f) for f in g["feats"]]
    encoder = torch.from_numpy(g["encoder"]).to(dev) if g.get("encoder") is not None else None
    D = feats[0].shape[-1]
    out, F, d, st = gsbp_amd.create_feature_field(t["means"], t["quats"], t["scales"], t["opac"], t["vms"], t["K"], W, H,
                                                   lambda v: feats[v].to(dev), D, encoder=encoder, return_partials=True)
    eng = gsbp_amd.Engine(t["means"].shape[0], W, H, device=dev)
    proj = eng.project(eng.view(t["vms"][0].cpu(), t["K"].cpu(), W, H), t["means"], t["quats"], t["scales"], t["opac"],
                       want_outputs=True)
    rep = capture_report(cap, out.cpu().numpy(), F.cpu().numpy(), d.cpu().numpy(),
                         *[proj[k].cpu().numpy() for k in ("radii", "means2d", "conics", "depths")])
    print("HIP vs gsplat capture", fname, rep)
    for k in ("F", "d", "out"):
        assert rep[k]["p99"] <= 1e-4, (k, rep[k])
        assert rep[k]["rows_over_1e-4"] <= max(1, int(0.002 * rep[k]["rows"])) and rep[k]["max"] <= 1e-2, (k, rep[k])


def _token_capture_cases():
    from util import capture_tool
    return capture_tool().TOKEN_CASES


@pytest.mark.parametrize("fname,cfgname,dim,grid", _token_capture_cases(), ids=[c[0] for c in _token_capture_cases()])
def test_hip_token_space_against_gsplat_capture(dev, fname, cfgname, dim, grid):
    """The token-space kernels (round 6) against a capture of REAL gsplat 1.4.0 running the reference's dino loop
    (backproject.py:242-289), when one has been committed; skipped otherwise (parity unpinned)."""
    from util import capture_report, capture_tool
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fname)
    if not os.path.exists(path):
        pytest.skip(f"{fname} not captured yet (needs CUDA + gsplat==1.4.0)")
    cap = dict(np.load(path))
    cfg = syn.CONFIGS[cfgname]
    g = capture_tool().token_case_inputs(cfgname, dim, grid)
    t = {k: torch.from_numpy(np.asarray(g[k])).to(dev) for k in ("means", "quats", "scales", "opac", "K", "vms")}
    toks = [torch.from_numpy(f).to(dev) for f in g["feats"]]
    assert gsbp_amd.Engine.can_scatter_tokens(toks[0], cfg.height, cfg.width)
    out, F, d, st = gsbp_amd.create_feature_field(t["means"], t["quats"], t["scales"], t["opac"], t["vms"], t["K"], cfg.width,
                                                   cfg.height, lambda v: toks[v], dim, reduction="mean", upsample="nearest",
                                                   return_partials=True)
    rep = capture_report(cap, out.cpu().numpy(), F.cpu().numpy(), d.cpu().numpy())
    print("HIP (token space) vs gsplat capture", fname, rep)
    for k in ("F", "d", "out"):
        assert rep[k]["p99"] <= 1e-4, (k, rep[k])
        assert rep[k]["rows_over_1e-4"] <= max(1, int(0.002 * rep[k]["rows"])) and rep[k]["max"] <= 1e-2, (k, rep[k])


def test_view_per_stream_schedule_waits_for_late_maps(dev):
    """Stream discipline of the view-per-stream schedule (small scene, <= 32-channel maps -> `independent`): (1) with an
    encoder, the encode kernel must run on the encoder stream behind the `ready` event even though engine 0 is bound to
    a view's stream; (2) without one, a feature function that produces its map LATE on the caller's stream (a long
    kernel in front of it) must still be consumed correctly (default feature_fn_stream_safe=False: one event per view).  Compared with serial single-stream runs of the same job."""
    cfg, sc = scene_np("T1", n_views=6)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=6).to(dev)
    K_in, n_out = 512, 16
    g = torch.Generator().manual_seed(5)
    wide = [torch.randn(cfg.height, cfg.width, K_in, generator=g).to(dev) for _ in range(6)]
    enc = (torch.randn(K_in, n_out, generator=g) / K_in ** 0.5).to(dev)
    ballast = torch.randn(4096, 4096, device=dev)

    def late(v):  # ~ms of work on the caller's stream in front of the map
        for _ in range(3):
            ballast @ ballast
        return wide[v].clone()

    def late_small(v):
        for _ in range(3):
            ballast @ ballast
        return (wide[v] @ enc).contiguous()

    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height)
    ref_e = gsbp_amd.create_feature_field(*args, lambda v: wide[v], K_in, encoder=enc, pipeline=False, return_partials=True)
    ref_s = gsbp_amd.create_feature_field(*args, lambda v: (wide[v] @ enc).contiguous(), n_out, pipeline=False,
                                          return_partials=True)
    # (1) encoder one view ahead: only encode_ahead's `ready` event orders the encode kernel behind the late map
    got_e = gsbp_amd.create_feature_field(*args, late, K_in, encoder=enc, pipeline=4, return_partials=True)
    # (2) no encoder, maps produced late on the caller's stream
    got_s = gsbp_amd.create_feature_field(*args, late_small, n_out, pipeline=4, return_partials=True)
    for got, ref in ((got_e, ref_e), (got_s, ref_s)):
        assert got[3]["n_pairs"] == ref[3]["n_pairs"] and got[3]["overflow"] == 0
        assert rel_row_err(got[1].cpu().numpy(), ref[1].cpu().numpy()) <= 2e-5
        assert np.abs((got[2] - ref[2]).cpu().numpy()).max() <= 1e-5 * float(ref[2].max())


@pytest.mark.parametrize("D", [384, 32, 130])
def test_scatter_nearest_upsampled_lowres_map(orc, dev, D):
    """dino variant (backproject.py:242-249): patch tokens [h,w,D] -> F.interpolate(nearest) -> scatter.  The HIP path
    reads the low-resolution map through index maps; the oracle gets the materialised upsampled map."""
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    lh, lw = 13, 17
    low = torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(11))
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="nearest")[0]
    up = up.permute(1, 2, 0)  # [H,W,D] view of a [D,H,W] tensor, exactly what the reference scatters
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    eng.blend_weights(view)
    F1 = torch.zeros(cfg.n_gaussians, D, device=dev)
    d1 = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter(view, low.to(dev), F1, d1, upsample="nearest")
    F2 = torch.zeros_like(F1)
    d2 = torch.zeros_like(d1)
    eng.scatter(view, up.to(dev), F2, d2)  # channel-major strides: the element-wise staging path
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height,
                         np.ascontiguousarray(up.numpy()), Fr, dr)
    assert rel_row_err(F1.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(F2.cpu().numpy(), Fr) <= 1e-4
    assert torch.allclose(d1, d2, rtol=1e-5, atol=0)  # atomics: order of the per-tile partial sums differs
    with pytest.raises(gsbp_amd.GwbpError):
        eng.scatter(view, low.to(dev), F1, d1)  # a low-resolution map without upsample= is a shape error


@pytest.mark.parametrize("D", [64, 132, 256, 384, 512, 768, 1024, 1280, 1536, 2052])
def test_token_space_scatter_against_oracle(orc, dev, D):
    """Round 6: the dino variant in TOKEN space (backproject.py:242-289).  An 8 x 12 map at 200 x 136 has texels of 17 x 16.7
    pixels -- at least a tile -- so every tile sees at most 2 x 2 of them: Engine.blend_tokens leaves per-record token-quadrant
    weight sums, Engine.scatter_tokens applies them with one plain read-modify-write per F row -- any D % 4 == 0 from 64 up: one
    to four 256-channel chunks side by side in one pass (64 / 132: most of the wave masked off; 384: a half chunk), two passes
    (1280: a chunk that does not exist, 1536), three passes with a 4-channel tail (2052).  Against the oracle fed the materialised F.interpolate(mode="nearest") map,
    against the pixel-slab path, alpha map bit for bit with blend_weights, and -- no atomics -- bit-identical on a rerun."""
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    lh, lw = 8, 12
    assert gsbp_amd.Engine.token_geometry_ok(lh, lw, cfg.height, cfg.width)
    low = torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(5))
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="nearest")[0]
    up = np.ascontiguousarray(up.permute(1, 2, 0).numpy())
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    a_ref = eng.blend_weights(view, want_alphas=True)
    F0 = torch.zeros(cfg.n_gaussians, D, device=dev)
    d0 = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter(view, low.to(dev), F0, d0, upsample="nearest")
    st0 = eng.stats()
    res = []
    for _ in range(2):
        a_tok = eng.blend_tokens(view, lh, lw, want_alphas=True)
        F1 = torch.full((cfg.n_gaussians, D), 0.25, device=dev)  # accumulates INTO F and d
        d1 = torch.full((cfg.n_gaussians,), 0.5, device=dev)
        eng.scatter_tokens(view, low.to(dev), F1, d1, 2.0, 3.0)
        res.append((F1, d1))
    st1 = eng.stats()
    # (the counters of one projection add up over its blends: one blend_weights + two blend_tokens)
    assert st1["overflow"] == 0 and st1["n_pairs"] == 3 * st0["n_pairs"] and st1["n_headers"] == 3 * st0["n_headers"]
    assert st1["blend_kind"] == 3
    assert torch.equal(a_tok, a_ref)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])  # deterministic: no atomics
    F2 = torch.full((cfg.n_gaussians, D), 0.25, device=dev)
    eng.scatter_tokens(view, low.to(dev), F2, None, 2.0, 3.0)  # d is optional (the sums stay in the workspace until the next blend)
    assert torch.equal(F2, res[0][0])
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height, up, Fr, dr)
    F1n, d1n = (res[0][0].cpu().numpy() - 0.25) / 2.0, (res[0][1].cpu().numpy() - 0.5) / 3.0
    touched = dr > 0
    assert rel_row_err(F1n, Fr) <= 1e-4
    assert rel_row_err(d1n[:, None], dr[:, None]) <= 1e-4
    assert np.array_equal(res[0][1].cpu().numpy() != 0.5, touched)  # exactly the Gaussians that receive weight are written
    assert rel_row_err(F0.cpu().numpy(), Fr) <= 1e-4
    with pytest.raises(gsbp_amd.GwbpError):
        eng.scatter(view, low.to(dev), F0, d0, upsample="nearest")  # no weight store behind blend_tokens
    with pytest.raises(gsbp_amd.GwbpError):
        eng.blend_tokens(view, 13, 17)  # texels narrower than a tile


def test_token_space_edge_cases(orc, dev):
    """Token space at the edges the reference's tests would hold if it had any: no Gaussians; an image smaller than a tile with a
    1 x 1 token map (every pixel one token: F[g, :] = d[g] * token); whole-screen splats whose tile rectangles span the image
    (96 x 64 = 24 tiles each, more than one batch of 16 emit slots per Gaussian) over a 3 x 5 token map; every Gaussian culled."""
    # N = 0
    eng0 = gsbp_amd.Engine(0, 40, 24, device=dev)
    z3, z4, z1 = torch.zeros(0, 3, device=dev), torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev)
    view0 = eng0.view(torch.eye(4), torch.tensor([[50.0, 0, 20], [0, 50.0, 12], [0, 0, 1]]), 40, 24)
    eng0.project(view0, z3, z4, z3, z1)
    eng0.bin_sort(view0)
    eng0.blend_tokens(view0, 1, 2)
    eng0.scatter_tokens(view0, torch.rand(1, 2, 256, device=dev), torch.zeros(0, 256, device=dev), torch.zeros(0, device=dev))
    assert eng0.stats()["n_pairs"] == 0 and eng0.stats()["overflow"] == 0
    g = torch.Generator().manual_seed(8)
    vm = torch.eye(4)
    for (W, H, lh, lw, n_big, n_small, D) in ((13, 9, 1, 1, 2, 20, 256), (96, 64, 3, 5, 6, 40, 512)):
        n = n_big + n_small
        K = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]])
        means = torch.cat([torch.tensor([[0.0, 0.0, 2.0 + 0.1 * i] for i in range(n_big)]),
                           torch.rand(n_small, 3, generator=g) * torch.tensor([1.6, 1.0, 1.0]) + torch.tensor([-0.8, -0.5, 1.5])])
        scales = torch.cat([torch.full((n_big, 3), 3.0), torch.full((n_small, 3), 0.05)])
        quats = torch.randn(n, 4, generator=g)
        opac = torch.cat([torch.full((n_big,), 0.05), torch.rand(n_small, generator=g)])
        low = torch.randn(lh, lw, D, generator=g)
        up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), mode="nearest")[0].permute(1, 2, 0)
        assert gsbp_amd.Engine.token_geometry_ok(lh, lw, H, W)
        eng = gsbp_amd.Engine(n, W, H, device=dev)
        view = eng.view(vm, K, W, H)
        eng.project(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
        eng.bin_sort(view)
        eng.blend_tokens(view, lh, lw)
        F, dd = torch.zeros(n, D, device=dev), torch.zeros(n, device=dev)
        eng.scatter_tokens(view, low.to(dev), F, dd)
        Fr, dr = np.zeros((n, D)), np.zeros(n)
        info = orc.backproject_view(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(), vm.numpy(), K.numpy(), W, H,
                                    np.ascontiguousarray(up.numpy()), Fr, dr)
        st = eng.stats()
        assert st["overflow"] == 0 and st["n_pairs"] == info["n_pairs"] > 0
        assert rel_row_err(F.cpu().numpy(), Fr) <= TOL and rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
        if lh == lw == 1:  # one token for the whole image: F[g, :] = d[g] * token
            assert float((F - dd[:, None] * low.to(dev)[0, 0][None]).abs().max()) <= 1e-5 * float(dd.max()) * float(low.abs().max())
    # every Gaussian behind the camera: nothing is touched
    eng.project(view, (means * torch.tensor([1.0, 1.0, -1.0])).to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
    eng.bin_sort(view)
    eng.blend_tokens(view, lh, lw)
    F2, d2 = torch.full((n, D), 7.0, device=dev), torch.full((n,), 3.0, device=dev)
    eng.scatter_tokens(view, low.to(dev), F2, d2)
    assert eng.stats()["n_visible"] == 0 and bool((F2 == 7.0).all()) and bool((d2 == 3.0).all())


def test_token_space_precondition_is_enforced_on_the_device(dev):
    """The C ABI's own guard: index maps that send a tile to more than 2 x 2 texels raise gwbp_stats.overflow bit 3 (the Python
    host never calls gwbp_blend_tokens with such maps; a foreign caller of the C ABI might)."""
    import ctypes as C
    from gsbp_amd._lib import ptr
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    ymap, xmap = eng.nearest_maps(13, 17, cfg.height, cfg.width)  # 10.5 x 11.8 pixel texels
    eng._call("gwbp_blend_tokens", *eng._args(), C.byref(view), ptr(ymap), ptr(xmap), None, eng._stream())
    assert eng.stats()["overflow"] & 8


def test_token_space_takes_views_up_to_4096_pixels_and_says_so(dev):
    """gwbp_scatter_tokens keeps 256-entry tile-column / -row tables in LDS: a 4112-pixel-wide view is refused with
    GWBP_EUNSUPPORTED by the C ABI (not run wrongly), Engine.can_scatter_tokens sends such a view to scatter(upsample="nearest")."""
    import ctypes as C
    from gsbp_amd._lib import ptr
    W, H, D = 4112, 32, 256
    tokens = torch.rand(1, 4, D, device=dev)
    assert gsbp_amd.Engine.can_scatter_tokens(tokens, 32, 4096)
    assert not gsbp_amd.Engine.can_scatter_tokens(tokens, H, W)
    eng = gsbp_amd.Engine(4, W, H, device=dev)
    K = torch.tensor([[3000.0, 0, W / 2], [0, 3000.0, H / 2], [0, 0, 1]])
    view = eng.view(torch.eye(4), K, W, H)
    means = torch.tensor([[0.0, 0, 4], [0.5, 0, 4], [-0.5, 0, 4], [0.2, 0, 5]], device=dev)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 4, device=dev)
    eng.project(view, means, quats, torch.full((4, 3), 0.05, device=dev), torch.full((4,), 0.8, device=dev))
    eng.bin_sort(view)
    eng.blend_tokens(view, 1, 4)
    F, d = torch.zeros(4, D, device=dev), torch.zeros(4, device=dev)
    with pytest.raises(gsbp_amd.GwbpError, match="4096"):
        eng.scatter_tokens(view, tokens, F, d)
    ymap, xmap = eng.nearest_maps(1, 4, H, W)
    with pytest.raises(gsbp_amd.GwbpError, match="tile columns"):
        eng._call("gwbp_scatter_tokens", *eng._args(), C.byref(view), ptr(tokens), C.c_int64(4 * D), C.c_int64(D), D, ptr(ymap),
                  ptr(xmap), C.c_float(1.0), C.c_float(1.0), ptr(F), ptr(d), eng._stream())
    assert float(F.abs().sum()) == 0.0


def test_token_space_non_finite_token_reaches_exactly_its_gaussians(orc, dev):
    """A NaN token (backproject.py:239-241 can produce one) must reach exactly the Gaussians that have weight inside it."""
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    lh, lw, D = 8, 12, 256
    low = torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(6))
    low[3, 5, 17] = float("nan")
    low[6, 2, :] = float("inf")
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="nearest")[0]
    up = np.ascontiguousarray(up.permute(1, 2, 0).numpy())
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    eng.blend_tokens(view, lh, lw)
    F1 = torch.zeros(cfg.n_gaussians, D, device=dev)
    d1 = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter_tokens(view, low.to(dev), F1, d1)
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height, up, Fr, dr)
    bad, bad_ref = ~np.isfinite(F1.cpu().numpy()).all(1), ~np.isfinite(Fr).all(1)
    assert bad_ref.sum() > 0 and np.array_equal(bad, bad_ref), (int(bad.sum()), int(bad_ref.sum()))
    assert rel_row_err(F1.cpu().numpy()[~bad], Fr[~bad]) <= 1e-4


def test_create_feature_field_token_space_equals_pixel_slabs(dev):
    """The driver takes the token path by itself for a nearest-upsampled map whose texels cover a tile (pipelined and serial),
    and both equal the pixel-slab kernels."""
    cfg, sc = scene_np("T1", n_views=5)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=5).to(dev)
    D, lh, lw = 512, 8, 12
    lows = [torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(30 + v)).to(dev) for v in range(5)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height)
    kw = dict(feature_fn=lambda v: lows[v], dim=D, reduction="mean", upsample="nearest", return_partials=True)
    ref = gsbp_amd.create_feature_field(*args, **kw, token_space=False)
    for pipeline in (True, False):
        got = gsbp_amd.create_feature_field(*args, **kw, pipeline=pipeline)
        assert got[3]["overflow"] == 0 and got[3]["n_pairs"] == ref[3]["n_pairs"]
        assert rel_row_err(got[1].cpu().numpy(), ref[1].cpu().numpy()) <= 2e-5
        assert np.abs((got[2] - ref[2]).cpu().numpy()).max() <= 1e-5 * float(ref[2].max())
        assert rel_row_err(got[0].cpu().numpy(), ref[0].cpu().numpy()) <= 2e-5


def test_create_feature_field_upsample_matches_materialised(dev):
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    D, lh, lw = 128, 9, 11
    lows = [torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(20 + v)).to(dev) for v in range(3)]

    def full(v):
        t = torch.nn.functional.interpolate(lows[v].permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="nearest")
        return t[0].permute(1, 2, 0)

    args = (d["means"], d["quats"], d["scales"], d["opac"], d["vms"][:3], d["K"], cfg.width, cfg.height)
    a = gsbp_amd.create_feature_field(*args, feature_fn=lambda v: lows[v], dim=D, reduction="mean", upsample="nearest")
    b = gsbp_amd.create_feature_field(*args, feature_fn=full, dim=D, reduction="mean")
    c = gsbp_amd.create_feature_field(*args, feature_fn=lambda v: lows[v], dim=D, reduction="mean", upsample="nearest",
                                      pipeline=False)
    assert rel_row_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5
    assert rel_row_err(c.cpu().numpy(), b.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("name", ["T0", "T1", "C1"])
def test_tight_binning_changes_only_the_lists(name, dev):
    """GWBP_FLAG_TIGHT_BINNING drops (Gaussian, tile) pairs that cannot contribute: the weight store -- every
    (gaussian, pixel, w) triple -- and therefore F and d must be identical to the exact binning, bit for bit."""
    cfg, sc = scene_np(name)
    d = to_dev(sc, dev)
    exact = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    tight = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
    for v in range(min(cfg.n_views, 2)):
        out = []
        for eng in (exact, tight):
            view, _, _ = _front(eng, d, cfg, v, want=False)
            alphas = eng.blend_weights(view, want_alphas=True)
            st = eng.stats()
            gid, pix, w = eng.dump_pairs(view)
            key, ws = sort_pairs(gid.cpu().numpy(), pix.cpu().numpy(), w.cpu().numpy())
            out.append((st, key, ws, alphas.cpu().numpy()))
        (s0, k0, w0, a0), (s1, k1, w1, a1) = out
        assert s1["n_pairs"] == s0["n_pairs"] and s1["n_visible"] == s0["n_visible"] and s1["n_headers"] == s0["n_headers"]
        assert s1["n_isect"] < s0["n_isect"]
        assert np.array_equal(k0, k1) and np.array_equal(w0.view(np.uint32), w1.view(np.uint32))
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))


@pytest.mark.parametrize("min_pairs", [0.0, 1e9], ids=["stay_wide", "switch_to_narrow"])
def test_pipelined_driver_with_the_wide_scatter_kernel(dev, monkeypatch, min_pairs):
    """D = 256: the pipeline starts on the 256-channel scatter kernel with d added behind the blend on the side stream,
    then either keeps it or switches to the 128-channel kernel after two views (fronts already enqueued keep their own
    d bookkeeping).  Must equal the serial driver (fused per-view call, 128-channel kernel)."""
    monkeypatch.setattr(gsbp_amd.ViewPipeline, "WIDE_MIN_PAIRS_PER_RECORD", min_pairs)
    cfg, sc = scene_np("T1", n_views=6)
    d = to_dev(sc, dev)
    D = 256
    vms = syn.make_cameras(cfg, n_views=6).to(dev)
    feats_all = [syn.make_feature_map(cfg, v, dim=D).to(dev) for v in range(6)]
    res = []
    for pipeline in (True, False):
        out, F, dd, st = gsbp_amd.create_feature_field(d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"],
                                                        cfg.width, cfg.height, lambda v: feats_all[v], D,
                                                        pipeline=pipeline, return_partials=True)
        res.append((F.cpu().numpy(), dd.cpu().numpy(), st["n_pairs"], out.cpu().numpy()))
    assert res[0][2] == res[1][2]
    assert rel_row_err(res[0][0], res[1][0]) <= 1e-5
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-5 * res[1][1].max()
    assert rel_row_err(res[0][3], res[1][3]) <= 1e-5


def test_wide_kernel_selfcheck_and_the_fallback_to_the_narrow_kernel(dev, monkeypatch):
    """The run-time half of the 256-channel kernel's safety net: the product library passes the self-check;
    a library that fails it (simulated verdict) makes the pipeline warn and scatter with the 128-channel kernel -- same field."""
    from gsbp_amd import backproject as bp
    bp._WIDE_OK.clear()
    assert bp.wide_kernel_selfcheck(dev) is True and len(bp._WIDE_OK) == 1
    cfg, sc = scene_np("T1", n_views=4)
    d = to_dev(sc, dev)
    D = 256
    vms = syn.make_cameras(cfg, n_views=4).to(dev)
    feats_all = [syn.make_feature_map(cfg, v, dim=D).to(dev) for v in range(4)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: feats_all[v], D)
    pipe = gsbp_amd.ViewPipeline(cfg.n_gaussians, cfg.width, cfg.height, dev, scatter_dim=D)
    assert pipe.wide is True
    ref = gsbp_amd.create_feature_field(*args)
    for k in list(bp._WIDE_OK):
        monkeypatch.setitem(bp._WIDE_OK, k, False)
    with pytest.warns(RuntimeWarning, match="128-channel"):
        pipe = gsbp_amd.ViewPipeline(cfg.n_gaussians, cfg.width, cfg.height, dev, scatter_dim=D)
    assert pipe.wide is False
    with pytest.warns(RuntimeWarning, match="128-channel"):
        out = gsbp_amd.create_feature_field(*args)
    assert rel_row_err(out.cpu().numpy(), ref.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("pipeline", [True, False], ids=["pipelined", "serial"])
def test_driver_grows_the_workspace_on_overflow(dev, pipeline):
    """Capacities far too small for the scene: the driver notices the overflow flags at the end of the pass, enlarges
    the workspace and starts over; the result must equal the run with ample capacities."""
    cfg, sc = scene_np("T1", n_views=4)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=4).to(dev)
    feats_all = [syn.make_feature_map(cfg, v).to(dev) for v in range(4)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: feats_all[v],
            cfg.feat_dim)
    small = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, isect_cap=3000, pair_cap=1 << 15,
                            tight_binning=True)
    a, Fa, da, st = gsbp_amd.create_feature_field(*args, engine=small, pipeline=pipeline, return_partials=True)
    assert small.isect_cap > 3000 and st["overflow"] == 0
    b, Fb, db, _ = gsbp_amd.create_feature_field(*args, pipeline=pipeline, return_partials=True)
    assert rel_row_err(Fa.cpu().numpy(), Fb.cpu().numpy()) <= 1e-5
    assert np.abs(da.cpu().numpy() - db.cpu().numpy()).max() <= 1e-5 * float(db.max())


@pytest.mark.parametrize("pipeline", [True, False], ids=["pipelined", "serial"])
def test_token_space_driver_survives_an_intersection_overflow(dev, pipeline):
    """The token-space path files its weight sums at EMIT positions; when the intersection capacity overflows there are none
    (k_emit returns at once).  gwbp_scatter_tokens must then touch nothing (it used to be able to read through a stale
    estart[]), the driver grows the workspace and the rerun equals a run with ample capacities."""
    cfg, sc = scene_np("T1", n_views=3)
    d = to_dev(sc, dev)
    vms = syn.make_cameras(cfg, n_views=3).to(dev)
    D, lh, lw = 256, 8, 12
    lows = [torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(60 + v)).to(dev) for v in range(3)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height)
    kw = dict(feature_fn=lambda v: lows[v], dim=D, reduction="mean", upsample="nearest", return_partials=True, pipeline=pipeline)
    small = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, isect_cap=3000, tight_binning=True)
    a, Fa, da, st = gsbp_amd.create_feature_field(*args, **kw, engine=small)
    assert small.isect_cap > 3000 and st["overflow"] == 0
    b, Fb, db, _ = gsbp_amd.create_feature_field(*args, **kw)
    assert rel_row_err(Fa.cpu().numpy(), Fb.cpu().numpy()) <= 1e-5
    assert np.abs(da.cpu().numpy() - db.cpu().numpy()).max() <= 1e-5 * float(db.max())


@pytest.mark.parametrize("D", [128, 256, 512])
def test_create_feature_field_bilinear_matches_materialised(D, dev):
    """The driver with upsample="bilinear" (low-resolution maps handed over) against the driver fed with
    F.interpolate(mode="bilinear", align_corners=False)'s materialised maps (backproject.py:108-113).  D % 256 == 0: the
    bilinear slab staging of the 256-channel kernel; 128: that of the 128-channel kernel."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    lh, lw = 12, 17
    lows = [torch.nn.functional.normalize(torch.randn(lh, lw, D, generator=torch.Generator().manual_seed(40 + v)), dim=2).to(dev)
            for v in range(3)]

    def full(v):
        t = torch.nn.functional.interpolate(lows[v].permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="bilinear",
                                            align_corners=False)
        return t[0].permute(1, 2, 0)

    args = (d["means"], d["quats"], d["scales"], d["opac"], d["vms"][:3], d["K"], cfg.width, cfg.height)
    a = gsbp_amd.create_feature_field(*args, feature_fn=lambda v: lows[v], dim=D, upsample="bilinear")
    b = gsbp_amd.create_feature_field(*args, feature_fn=full, dim=D)
    c = gsbp_amd.create_feature_field(*args, feature_fn=lambda v: lows[v], dim=D, upsample="bilinear", pipeline=False)
    assert rel_row_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5
    assert rel_row_err(c.cpu().numpy(), b.cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("H,W,K,n", [(37, 53, 512, 16), (16, 16, 64, 5), (9, 7, 32, 1), (40, 24, 128, 13)])
def test_encode_map_matches_matmul(dev, H, W, K, n):
    """gwbp_encode_map (backproject_compressed.py:127, fp32 MFMA skinny GEMM) against a float64 matmul; dense and
    pixel-strided maps, pixel counts that are not multiples of the 16-pixel MFMA tile, fewer than 16 outputs."""
    g = torch.Generator().manual_seed(H * 1000 + W)
    feats = torch.randn(H, W, K, generator=g)
    enc = torch.randn(K, n, generator=g) / K ** 0.5
    eng = gsbp_amd.Engine(16, W, H, device=dev)
    ref = (feats.double() @ enc.double()).numpy()
    out = eng.encode_map(feats.to(dev), enc.to(dev))
    assert tuple(out.shape) == (H, W, n)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * K ** 0.5
    # a view into a wider buffer: pixel stride 2K, row stride with a gap
    big = torch.zeros(H, W + 3, 2 * K, device=dev)
    big[:, :W, :K] = feats.to(dev)
    out2 = eng.encode_map(big[:, :W, :K], enc.to(dev))
    assert torch.equal(out2, out)
    # shapes the kernel does not take RAISE (no silent library-GEMM fallback on the hot stage)
    assert not gsbp_amd.Engine.can_encode_map(feats.to(dev)[:, :, : K - 8], enc.to(dev)[: K - 8])
    with pytest.raises(gsbp_amd.GwbpError):
        eng.encode_map(feats.to(dev)[:, :, : K - 8], enc.to(dev)[: K - 8])
    # an explicit stream: launched there (not on the engine's bound / current stream), K = 2048 after a smaller K (the
    # dynamic-LDS limit must follow the larger request)
    st = torch.cuda.Stream(device=dev)
    big_k = torch.randn(8, 8, 2048, generator=g)
    enc_k = torch.randn(2048, n, generator=g) / 2048 ** 0.5
    bk, ek = big_k.to(dev), enc_k.to(dev)
    torch.cuda.synchronize(dev)
    out4 = eng.encode_map(bk, ek, stream=st)
    st.synchronize()
    ref4 = (big_k.double() @ enc_k.double()).numpy()
    assert np.abs(out4.cpu().numpy() - ref4).max() <= 2e-6 * max(1.0, np.abs(ref4).max()) * 2048 ** 0.5


@pytest.mark.parametrize("K,n", [(512, 16), (64, 5), (32, 16)])
def test_scatter_encoded_matches_scatter_of_encoded_map(orc, dev, K, n):
    """gwbp_scatter_encoded (the encoder of backproject_compressed.py:127 fused into the slab staging) against the plain
    scatter of the materialised feats @ encoder, and against the oracle fed the CPU-encoded map; image size not a
    multiple of the tile (edge tiles), fewer than 16 outputs, a pixel-strided map."""
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    g = torch.Generator().manual_seed(K + n)
    feats = torch.randn(cfg.height, cfg.width, K, generator=g)
    enc = torch.randn(K, n, generator=g) / K ** 0.5
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    view = eng.view(d["vms"][0], d["K"], cfg.width, cfg.height)
    eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
    eng.bin_sort(view)
    eng.blend_weights(view)
    big = torch.zeros(cfg.height, cfg.width + 2, K + 16, device=dev)
    big[:, :cfg.width, :K] = feats.to(dev)
    res = []
    for fmap in (feats.to(dev), big[:, :cfg.width, :K]):
        F = torch.zeros(cfg.n_gaussians, n, device=dev)
        dd = torch.zeros(cfg.n_gaussians, device=dev)
        eng.scatter_encoded(view, fmap, enc.to(dev), F, dd)
        res.append((F, dd))
    assert torch.equal(res[0][0], res[1][0]) or float((res[0][0] - res[1][0]).abs().max()) <= 1e-5 * float(res[0][0].abs().max())
    F2 = torch.zeros(cfg.n_gaussians, n, device=dev)
    d2 = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter(view, eng.encode_map(feats.to(dev), enc.to(dev)), F2, d2)
    scale = float(F2.norm(dim=1).max())
    assert float((res[0][0] - F2).norm(dim=1).max()) <= 2e-5 * scale
    assert float((res[0][1] - d2).abs().max()) <= 2e-5 * float(d2.max())
    Fr = np.zeros((cfg.n_gaussians, n), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height,
                         (feats @ enc).numpy(), Fr, dr)
    assert rel_row_err(res[0][0].cpu().numpy(), Fr) <= TOL
    assert rel_row_err(res[0][1].cpu().numpy()[:, None], dr[:, None]) <= TOL


@pytest.mark.parametrize("split", [False, True], ids=["one_wave_per_tile", "producer_consumer"])
@pytest.mark.parametrize("name,K,n", [("T1", 512, 16), ("T1", 64, 5), ("T0", 32, 16), ("C1", 128, 16)])
def test_blend_scatter_encoded_matches_encode_then_blend_scatter(orc, dev, name, K, n, split):
    """gwbp_blend_scatter_encoded (round 5: the 512 -> 16 encoder of backproject_compressed.py:127 inside the fused blend +
    scatter kernel's tile prologue, on the matrix cores) against gwbp_encode_map + gwbp_blend_scatter of the same view -- the
    encoded pixels are the same k-ordered fp32 chain, so the weights are equal and F, d differ by summation order only -- and
    against the oracle fed the CPU-encoded map.  Edge tiles (image size not a multiple of 16), fewer than 16 outputs, a
    pixel-strided map; the alpha maps must be equal bit for bit."""
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    g = torch.Generator().manual_seed(K + n)
    feats = torch.randn(cfg.height, cfg.width, K, generator=g)
    enc = torch.randn(K, n, generator=g) / K ** 0.5
    N = cfg.n_gaussians
    eng = gsbp_amd.Engine(N, cfg.width, cfg.height, device=dev)
    # round 6: the same entry point in its producer / consumer form (GWBP_FLAG_SPLIT_ENCODER: encoder waves fill an LDS ring of
    # encoded tiles, blend waves drain it, one persistent workgroup per CU) -- same encoded pixels, same weights
    eng.set_split_encoder(split)
    view = eng.view(d["vms"][0], d["K"], cfg.width, cfg.height)

    def front():
        eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
        eng.bin_sort(view)

    big = torch.zeros(cfg.height, cfg.width + 2, K + 16, device=dev)
    big[:, :cfg.width, :K] = feats.to(dev)
    res = []
    for fmap in (feats.to(dev), big[:, :cfg.width, :K]):
        F, dd = torch.zeros(N, n, device=dev), torch.zeros(N, device=dev)
        front()
        alphas = eng.blend_scatter_encoded(view, fmap, enc.to(dev), F, dd, want_alphas=True)
        assert eng.stats()["overflow"] == 0
        res.append((F, dd, alphas))
    scale = float(res[0][0].norm(dim=1).max())
    assert float((res[0][0] - res[1][0]).norm(dim=1).max()) <= 2e-5 * scale and torch.equal(res[0][2], res[1][2])
    F2, d2 = torch.zeros(N, n, device=dev), torch.zeros(N, device=dev)
    front()
    a2 = eng.blend_scatter(view, eng.encode_map(feats.to(dev), enc.to(dev)), F2, d2, want_alphas=True)
    assert torch.equal(res[0][2], a2)
    assert float((res[0][0] - F2).norm(dim=1).max()) <= 2e-5 * scale
    assert float((res[0][1] - d2).abs().max()) <= 2e-5 * float(d2.max())
    Fr, dr = np.zeros((N, n), np.float64), np.zeros(N, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height,
                         (feats @ enc).numpy(), Fr, dr)
    assert rel_row_err(res[0][0].cpu().numpy(), Fr) <= TOL
    assert rel_row_err(res[0][1].cpu().numpy()[:, None], dr[:, None]) <= TOL
    with pytest.raises(gsbp_amd.GwbpError):  # K beyond the 32 KB the encoder may take in LDS
        eng.blend_scatter_encoded(view, torch.zeros(cfg.height, cfg.width, 1024, device=dev), torch.zeros(1024, 16, device=dev),
                                  torch.zeros(N, 16, device=dev), None)


def test_c_abi_refuses_wide_scatter_of_a_narrow_blend_and_leaves_f_and_d_untouched(dev):
    """include/gwbp.h, gwbp_stats.overflow bit 2: gwbp_scatter asked for the 256-channel kernel (D % 256 == 0, no
    GWBP_FLAG_NARROW_SCATTER) on a view that was blended WITH the flag finds no weight sums in the headers: the call raises bit 2
    and leaves BOTH F and d untouched (k_accum_d and the scatter kernel check the same thing), so that the documented recovery
    -- scatter again with the flag set -- counts nothing twice.  Engine.scatter guards this in Python; here the guard is
    bypassed and the C ABI is asked directly."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    D = 256
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    eng.set_narrow_scatter(True)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    eng.blend_weights(view)
    assert eng.stats()["blend_kind"] == 0
    feats = syn.make_feature_map(cfg, 0, dim=D).to(dev)
    F = torch.zeros(cfg.n_gaussians, D, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    eng.set_narrow_scatter(False)
    eng._halves = True  # lie to the Python-side guard: the library must catch it on its own
    eng.scatter(view, feats, F, dd)
    st = eng.stats()
    assert st["overflow"] & 4
    assert float(F.abs().max()) == 0.0 and float(dd.abs().max()) == 0.0
    # the documented recovery: the same view through the 128-channel kernel, once
    eng._halves = False
    eng.set_narrow_scatter(True)
    eng.scatter(view, feats, F, dd)
    F2 = torch.zeros_like(F)
    d2 = torch.zeros_like(dd)
    view, _, _ = _front(eng, d, cfg, 0, want=False)
    eng.blend_weights(view)
    eng.scatter(view, feats, F2, d2)
    assert float(dd.sum()) > 0
    scale = float(F2.norm(dim=1).max())
    assert float((F - F2).norm(dim=1).max()) <= 2e-5 * scale and float((dd - d2).abs().max()) <= 2e-5 * float(d2.max())


def test_dropin_denominator_pass_from_the_weight_sums(orc, dev):
    """The reference's second pass per view, `rasterization(zeros [N,3])[0][0].sum().backward()` (backproject.py:133-147),
    through the drop-in right after a 256-channel pass of the same view: the render of the zero table is not computed (its
    alphas are the blend's), and the backward of the sum -- one value expanded over [H,W,3] -- is that value times the
    per-record weight sums the view already holds (Engine.scatter_uniform) instead of a scatter of a map of ones.  Both
    against the explicit kernels, and the denominators against the oracle."""
    from gsbp_amd import rasterization
    from gsbp_amd.rasterization import get_engine
    cfg, sc = scene_np("T1")
    d, h = to_dev(sc, dev), npy(sc)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    args = (d["means"], d["quats"], d["scales"], d["opac"])
    wide = torch.zeros(N, 256, device=dev, requires_grad=True)
    table0 = torch.zeros(N, 3, device=dev, requires_grad=True)
    out, _, _ = rasterization(*args, wide, d["vms"][0][None], d["K"][None], width=W, height=H)
    (out[0] * syn.make_feature_map(cfg, 0, dim=256).to(dev)).sum().backward()
    eng = get_engine(dev, N, W, H)
    gen = eng.generation
    out0, alpha0, _ = rasterization(*args, table0, d["vms"][0][None], d["K"][None], width=W, height=H)
    assert eng.generation == gen and eng.has_weight_sums()  # the first pass's front stage, blended for the 256-channel kernel
    assert out0.shape == (1, H, W, 3) and float(out0.abs().max()) == 0.0
    (out0[0].sum() * 2.5).backward()  # (2.5: the expanded value is read, not assumed to be one)
    got = table0.grad.clone()
    # the explicit kernels on the same view: alphas of the pixel rasteriser, scatter of a materialised constant map
    with torch.no_grad():
        rgb = torch.rand(N, 3, device=dev)
        _, alpha_px, _ = rasterization(*args, rgb, d["vms"][0][None], d["K"][None], width=W, height=H)
    assert torch.allclose(alpha0, alpha_px, rtol=0, atol=2e-6)
    eng2 = gsbp_amd.Engine(N, W, H, device=dev)
    view, _, _ = _front(eng2, d, cfg, 0, want=False)
    eng2.blend_weights(view)
    assert not eng2.has_weight_sums()
    with pytest.raises(gsbp_amd.GwbpError):
        eng2.scatter_uniform(view, torch.tensor(1.0, device=dev), torch.zeros(N, 3, device=dev))
    want = torch.zeros(N, 3, device=dev)
    eng2.scatter(view, torch.full((H, W, 3), 2.5, device=dev), want, None)
    scale = float(want.abs().max())
    assert scale > 0 and float((got - want).abs().max()) <= 2e-5 * scale
    assert torch.equal(got[:, 0], got[:, 1]) and torch.equal(got[:, 0], got[:, 2])
    Fr, dr = np.zeros((N, 3), np.float64), np.zeros(N, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], W, H,
                         np.full((H, W, 3), 2.5, np.float32), Fr, dr)
    assert dr.max() > 0
    assert rel_row_err(got.cpu().numpy(), Fr) <= TOL
    assert rel_row_err(got[:, :1].cpu().numpy() / 2.5, dr[:, None]) <= TOL


def _nonfinite_case(orc, dev, name, D, fused, enc_k=None, split=False):
    """One view, a feature map with a NaN pixel and a +inf pixel: NaN must reach exactly the Gaussians that have a weight at one
    of those pixels (the reference's backward adds fac * v_render only for contributing pairs), every other row must match the
    oracle.  backproject.py:109 produces such pixels: feats / feats.norm() of an all-zero pixel."""
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    g = torch.Generator().manual_seed(11)
    K = enc_k or D
    feats = torch.randn(H, W, K, generator=g)
    feats[H // 2, W // 2, :] = float("nan")
    feats[H // 3, W // 4, K // 2] = float("inf")
    enc = torch.randn(K, D, generator=g) / K ** 0.5 if enc_k else None
    eff = feats if enc is None else feats @ enc
    Fr, dr = np.zeros((N, D), np.float64), np.zeros(N, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], W, H, eff.numpy(), Fr, dr)
    bad_ref = ~np.isfinite(Fr).all(axis=1)
    assert 0 < bad_ref.sum() < N // 2
    eng = gsbp_amd.Engine(N, W, H, device=dev)
    eng.set_split_encoder(split)
    view = eng.view(d["vms"][0], d["K"], W, H)
    eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
    eng.bin_sort(view)
    F, dd = torch.zeros(N, D, device=dev), torch.zeros(N, device=dev)
    if enc is not None:
        eng.blend_scatter_encoded(view, feats.to(dev), enc.to(dev), F, dd)
    elif fused:
        eng.blend_scatter(view, feats.to(dev), F, dd)
    else:
        eng.blend_weights(view)
        eng.scatter(view, feats.to(dev), F, dd)
    Fh = F.cpu().numpy()
    bad = ~np.isfinite(Fh).all(axis=1)
    assert np.array_equal(bad, bad_ref), (int(bad.sum()), int(bad_ref.sum()))
    assert rel_row_err(Fh[~bad], Fr[~bad]) <= TOL
    assert rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL


@pytest.mark.parametrize("name,D,fused,enc_k,split",
                         [("T1", 8, False, None, False), ("T1", 130, False, None, False), ("T1", 128, False, None, False),
                          ("T1", 256, False, None, False), ("T1", 16, True, None, False), ("T1", 5, True, None, False),
                          ("C1", 32, True, None, False), ("T1", 16, False, 64, False), ("T1", 16, False, 64, True)],
                         ids=["small_D8", "general_D130", "narrow_D128", "wide_D256", "fused_D16", "fused_D5", "fused_quarter_D32",
                              "fused_encoder_64to16", "fused_encoder_64to16_producer_consumer"])
def test_non_finite_features_reach_exactly_the_gaussians_that_touch_them(orc, dev, name, D, fused, enc_k, split):
    _nonfinite_case(orc, dev, name, D, fused, enc_k, split)


def test_profile_build_of_the_wide_kernel_agrees_with_the_narrow_one(dev):
    """k_scatter_wide keeps asm-issued loads in flight across compiler-visible code; a build with a different register allocation
    (PROFILE + in-kernel stamps) once copied two landing registers in front of their wait.  The PROFILE library is built from
    the SAME sources as the product library (`_lib.build_profile()`: by __graft_entry__.build() in the container, by this test
    where hipcc exists; a library older than the sources FAILS instead of being tested) and the wide-vs-narrow comparison runs
    on it in a child process (one library per process)."""
    import subprocess
    import sys
    from gsbp_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _lib.PROFILE_LIB_PATH
    if os.path.exists("/opt/rocm/bin/hipcc"):
        _lib.build_profile()
    assert os.path.exists(lib), "tools/lib/libgwbp_profile.so missing: __graft_entry__.build() makes it"
    assert not _lib._stale(lib), "tools/lib/libgwbp_profile.so is older than csrc/: rebuild (python __graft_entry__.py)"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "wide_vs_narrow.py"), lib], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
