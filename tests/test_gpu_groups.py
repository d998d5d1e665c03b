"""The block-sparse scatter on the matrix cores (GWBP_FLAG_GROUP_SCATTER, csrc/scatter_mfma.hip): k_group_sort + k_pack
behind the blend, k_scatter_mfma as the scatter.  Same semantics as the vector kernels: F[g,:] += sum_p w_g(p) feats[p,:]
(backproject.py:127-131), every record's sum in ascending pixel order, exact fp32 (v_mfma_f32_16x16x4_f32 is a k-ordered
fmaf chain; a zero weight adds +-0).  Checked against the CPU oracle, against the 128-channel vector kernel on the SAME
weight store, on non-finite feature values (0 x NaN must not leak into records that do not touch the pixel) and at the
capacity edge."""
import numpy as np
import pytest
import torch

from util import npy, rel_row_err, scene_np, to_dev

import gsbp_amd
from gsbp_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _blend(eng, d, cfg, v):
    view = eng.view(d["vms"][v].cpu(), d["K"].cpu(), cfg.width, cfg.height)
    eng.project(view, d["means"], d["quats"], d["scales"], d["opac"])
    eng.bin_sort(view)
    eng.blend_weights(view)
    return view


@pytest.mark.parametrize("name,D", [("T1", 128), ("T1", 384), ("C1", 256), ("T0", 128)])
def test_group_scatter_matches_oracle_and_vector_kernel(orc, dev, name, D):
    cfg, sc = scene_np(name)
    d, h = to_dev(sc, dev), npy(sc)
    feats = torch.randn(cfg.height, cfg.width, D, generator=torch.Generator().manual_seed(D))
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True)
    view = _blend(eng, d, cfg, 0)
    st = eng.stats()
    assert st["overflow"] == 0 and st["blend_kind"] == 3  # store + record groups
    F = torch.zeros(cfg.n_gaussians, D, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter(view, feats.to(dev), F, dd)
    # the same weight store through the 128-channel vector kernel
    eng.set_group_scatter(False)
    F2 = torch.zeros_like(F)
    d2 = torch.zeros_like(dd)
    eng.scatter(view, feats.to(dev), F2, d2)
    assert eng.stats()["overflow"] == 0
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    orc.backproject_view(h["means"], h["quats"], h["scales"], h["opac"], h["vms"][0], h["K"], cfg.width, cfg.height,
                         feats.numpy(), Fr, dr)
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL and rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
    # per record both kernels run the same fmaf chain; only the order of the flush atomics differs
    assert rel_row_err(F.cpu().numpy(), F2.cpu().numpy().astype(np.float64)) <= 2e-6
    assert torch.allclose(dd, d2, rtol=1e-5, atol=0)
    # a second scatter of the same view (the queues re-arm themselves) and scale_f
    F3 = torch.zeros_like(F)
    eng.set_group_scatter(True)
    eng.scatter(view, feats.to(dev), F3, None, scale_f=0.5)
    assert rel_row_err(2.0 * F3.cpu().numpy(), F.cpu().numpy().astype(np.float64)) <= 2e-6


def test_non_finite_features_reach_exactly_the_gaussians_that_touch_them(dev):
    """feats / feats.norm() of an all-zero pixel is NaN in the reference (backproject.py:109): it reaches exactly the
    Gaussians with weight on that pixel.  A dense operand table multiplies every pixel of the group's union by every record
    (0 x NaN = NaN), so tiles whose staged pixels are not all finite take the exact sparse path; the vector kernels must
    not multiply a zero-weight padding lane by a non-finite pixel either (0 x inf).  Expectation: the plain sum over the
    view's (Gaussian, pixel, weight) triples in float64 -- NaN / +-inf patterns must agree exactly, finite values closely."""
    cfg, sc = scene_np("T1")
    d = to_dev(sc, dev)
    D = 128
    feats = torch.randn(cfg.height, cfg.width, D, generator=torch.Generator().manual_seed(3))
    feats[10, 20, :] = float("nan")
    feats[50, 100, 7] = float("inf")
    feats[100, 150, 64:] = float("-inf")
    feats[0, 0, 3] = float("nan")      # the first pixel of tile 0
    feats[31, 47, :] = float("inf")    # the last pixel of a tile
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True)
    view = _blend(eng, d, cfg, 0)
    gid, pix, w = [t.cpu().numpy() for t in eng.dump_pairs(view)]
    want = np.zeros((cfg.n_gaussians, D))
    with np.errstate(invalid="ignore"):
        np.add.at(want, gid, w[:, None].astype(np.float64) * feats.reshape(-1, D).numpy().astype(np.float64)[pix])
    assert np.isnan(want).any() and np.isinf(want).any()
    got = {}
    for name, setup in (("groups", lambda: eng.set_group_scatter(True)),
                        ("narrow", lambda: (eng.set_group_scatter(False), eng.set_narrow_scatter(True))),):
        setup()
        F = torch.zeros(cfg.n_gaussians, D, device=dev)
        eng.scatter(view, feats.to(dev), F, None)
        got[name] = F.cpu().numpy()
    # the 256-channel vector kernel on the same view (its own blend: it needs the half-tile lists)
    D2 = 256
    feats2 = torch.cat([feats, feats], dim=2).contiguous()
    wide = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    wide.set_narrow_scatter(False)
    view2 = _blend(wide, d, cfg, 0)
    F = torch.zeros(cfg.n_gaussians, D2, device=dev)
    wide.scatter(view2, feats2.to(dev), F, None)
    got["wide"] = F.cpu().numpy()[:, :D]
    both = F.cpu().numpy()  # (the two identical halves differ only by the order of the flush atomics)
    assert np.array_equal(np.isnan(both[:, :D]), np.isnan(both[:, D:]))
    # the generic kernel (D % 128 != 0, D > 64) on the same view: 136 channels = the map + its first 8 channels again
    feats3 = torch.cat([feats, feats[:, :, :8]], dim=2).contiguous()
    F = torch.zeros(cfg.n_gaussians, 136, device=dev)
    eng.scatter(view, feats3.to(dev), F, None)
    got["generic"] = F.cpu().numpy()[:, :D]
    for name, g in got.items():
        assert np.array_equal(np.isnan(g), np.isnan(want)), name
        assert np.array_equal(np.isposinf(g), np.isposinf(want)) and np.array_equal(np.isneginf(g), np.isneginf(want)), name
        fin = np.isfinite(want)
        assert np.abs(g[fin] - want[fin]).max() <= 1e-5 * np.abs(want[fin]).max(), name


def test_group_tables_overflow_is_flagged_and_grown(dev):
    """Block capacity follows pair_cap: a workspace with a pair_cap that just fits the weight store overflows the dense tables
    (gwbp_stats.overflow bit 3); create_feature_field grows and restarts, the result equals the roomy run's."""
    cfg, sc = scene_np("C1")
    d = to_dev(sc, dev)
    D = 128
    vms = syn.make_cameras(cfg).to(dev)
    maps = [torch.randn(cfg.height, cfg.width, D, generator=torch.Generator().manual_seed(v)).to(dev) for v in range(4)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: maps[v], D)
    ref, Fr, dr, st_r = gsbp_amd.create_feature_field(*args, return_partials=True, allow_groups=True)
    assert st_r["overflow"] == 0
    probe = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True)
    view = _blend(probe, d, cfg, 0)
    used = probe.stats()["n_pairs"]
    small = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True,
                            pair_cap=max(1 << 15, int(1.3 * used)), isect_cap=probe.isect_cap)
    _blend(small, d, cfg, 0)
    st = small.stats()
    if not st["overflow"] & 8:
        pytest.skip("the dense tables fit this pair_cap: nothing to grow")
    out, F, dd, st2 = gsbp_amd.create_feature_field(*args, engine=small, return_partials=True)
    assert st2["overflow"] == 0 and small.pair_cap > int(1.3 * used)
    assert rel_row_err(F.cpu().numpy(), Fr.cpu().numpy().astype(np.float64)) <= 2e-6


def test_pipelined_driver_uses_groups_and_matches_vector_kernels(dev):
    cfg, sc = scene_np("T1", n_views=5)
    d = to_dev(sc, dev)
    D = 256
    vms = syn.make_cameras(cfg, n_views=5).to(dev)
    maps = [torch.randn(cfg.height, cfg.width, D, generator=torch.Generator().manual_seed(v)).to(dev) for v in range(5)]
    args = (d["means"], d["quats"], d["scales"], d["opac"], vms, d["K"], cfg.width, cfg.height, lambda v: maps[v], D)
    res = []
    for kw in (dict(allow_groups=True), dict(), dict(allow_wide=False), dict(pipeline=False, allow_groups=True)):
        out, F, dd, st = gsbp_amd.create_feature_field(*args, return_partials=True, **kw)
        assert st["overflow"] == 0
        res.append((F.cpu().numpy().astype(np.float64), dd.cpu().numpy(), st["n_pairs"]))
    for r in res[1:]:
        assert r[2] == res[0][2]
        assert rel_row_err(res[0][0], r[0]) <= 1e-5 and np.abs(res[0][1] - r[1]).max() <= 1e-4 * r[1].max()


def test_a_tile_with_more_records_than_one_sort_segment(orc, dev):
    """k_group_sort orders a tile's records in segments of 4096: 12 000 small, faint Gaussians crowded around the image
    centre put well over 4096 contributing records into a few tiles (no pixel terminates).  Groups path vs oracle vs the
    128-channel vector kernel; also exercises heavy tiles in the blend's page allocation."""
    cfg = syn.Config("HEAVY", 12_000, 1, 200, 136, 128, 0.01, False)
    g = torch.Generator().manual_seed(21)
    vm = syn.make_cameras(cfg)[0]
    R, t = vm[:3, :3], vm[:3, 3]
    # camera-space points in a 0.2 x 0.2 patch on the optical axis at the distance of the origin, back to world space
    pc = torch.stack([(torch.rand(cfg.n_gaussians, generator=g) - 0.5) * 0.2,
                      (torch.rand(cfg.n_gaussians, generator=g) - 0.5) * 0.2,
                      float(t[2]) + (torch.rand(cfg.n_gaussians, generator=g) - 0.5) * 0.2], dim=1)
    means = ((pc - t) @ R).contiguous()  # R^T (pc - t)
    quats = torch.randn(cfg.n_gaussians, 4, generator=g)
    scales = torch.full((cfg.n_gaussians, 3), 0.008)
    opac = torch.full((cfg.n_gaussians,), 0.03)
    K = syn.intrinsics(cfg)
    D = cfg.feat_dim
    feats = torch.randn(cfg.height, cfg.width, D, generator=g)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True)
    view = eng.view(vm, K, cfg.width, cfg.height)
    dm, dq, ds, do = [x.to(dev) for x in (means, quats, scales, opac)]
    while True:
        eng.project(view, dm, dq, ds, do)
        eng.bin_sort(view)
        eng.blend_weights(view)
        st = eng.stats()
        if not st["overflow"]:
            break
        eng.grow(st)
    gid, pix, w = [x.cpu().numpy() for x in eng.dump_pairs(view)]
    tile = (pix // cfg.width // 16) * 13 + (pix % cfg.width) // 16
    per_tile = np.bincount(np.unique(tile.astype(np.int64) * cfg.n_gaussians + gid) // cfg.n_gaussians)
    assert per_tile.max() > 4096, per_tile.max()  # the point of the test
    F = torch.zeros(cfg.n_gaussians, D, device=dev)
    dd = torch.zeros(cfg.n_gaussians, device=dev)
    eng.scatter(view, feats.to(dev), F, dd)
    eng.set_group_scatter(False)
    F2 = torch.zeros_like(F)
    d2 = torch.zeros_like(dd)
    eng.scatter(view, feats.to(dev), F2, d2)
    assert eng.stats()["overflow"] == 0
    Fr = np.zeros((cfg.n_gaussians, D), np.float64)
    dr = np.zeros(cfg.n_gaussians, np.float64)
    info = orc.backproject_view(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(), vm.numpy(), K.numpy(),
                                cfg.width, cfg.height, feats.numpy(), Fr, dr)
    assert info["n_pairs"] == st["n_pairs"]
    assert rel_row_err(F.cpu().numpy(), Fr) <= TOL and rel_row_err(dd.cpu().numpy()[:, None], dr[:, None]) <= TOL
    assert rel_row_err(F.cpu().numpy(), F2.cpu().numpy().astype(np.float64)) <= 2e-6
