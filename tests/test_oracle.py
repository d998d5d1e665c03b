"""CPU tests of the oracle (oracle/gwbp_oracle.c): golden vectors, an independent float64 formulation, the
literal autograd formulation of backproject.py, and closed-form known-answer cases.  No GPU needed."""
import math
import os

import numpy as np
import pytest
import torch

from util import rel_row_err

import gsbp_amd  # noqa: F401
from gsbp_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g0.npz")
W0, H0 = 64, 48


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def test_exp_neg_accuracy_and_golden(orc, gold):
    for x, y in zip(gold["exp_x"], gold["exp_y"]):
        v = orc.exp_neg(float(x))
        assert np.float32(v).view(np.uint32) == np.float32(y).view(np.uint32)  # deterministic bit pattern
        assert abs(v - math.exp(float(x))) <= 3e-7 * math.exp(float(x))
    assert orc.exp_neg(0.0) == 1.0


def test_oracle_reproduces_golden_vectors(orc, gold):
    g = gold
    proj = orc.project(g["means"], g["quats"], g["scales"], g["vms"][0], g["K"], W0, H0)
    for k in ("radii", "means2d", "conics", "depths"):
        assert np.array_equal(proj[k].view(np.uint32), g["v0_" + k].view(np.uint32)), k
    bins = orc.bin_sort(proj, W0, H0)
    assert np.array_equal(bins["isect_ids"], g["v0_isect_ids"])
    assert np.array_equal(bins["flatten_ids"], g["v0_flatten_ids"])
    assert np.array_equal(bins["tile_offsets"], g["v0_tile_offsets"])
    gid, pix, w, alphas = orc.blend_pairs(proj, bins, g["opac"], W0, H0, want_alphas=True)
    assert np.array_equal(gid, g["v0_pair_gid"]) and np.array_equal(pix, g["v0_pair_pix"])
    assert np.array_equal(w.view(np.uint32), g["v0_pair_w"].view(np.uint32))
    assert np.array_equal(alphas.view(np.uint32), g["v0_alphas"].view(np.uint32))
    out, F, d, stats = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W0,
                                              H0, lambda v: g["feats"][v], 8)
    assert [s["n_pairs"] for s in stats] == list(g["n_pairs"]) and [s["n_isect"] for s in stats] == list(g["n_isect"])
    assert np.array_equal(F, g["F"]) and np.array_equal(d, g["d"])  # double sums in a fixed order: exact
    assert np.array_equal(out.view(np.uint32), g["out"].view(np.uint32))


def test_float_accumulators_and_threads_agree(orc, gold):
    g = gold
    res = []
    for acc, nt in ((np.float64, 1), (np.float32, 1), (np.float32, 4)):
        out, F, d, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W0, H0,
                                              lambda v: g["feats"][v], 8, acc=acc, nthreads=nt)
        res.append((out, F))
    assert rel_row_err(res[1][1], res[0][1]) < 2e-6
    assert np.array_equal(res[1][1], res[2][1])  # channel-sliced threading: bitwise reproducible
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-6


def test_oracle_matches_independent_float64_formulation(orc, gold):
    import ref_np
    g = gold
    proj64 = ref_np.project(g["means"], g["quats"], g["scales"], g["vms"][0], g["K"], W0, H0)
    proj = orc.project(g["means"], g["quats"], g["scales"], g["vms"][0], g["K"], W0, H0)
    ok = proj["radii"] > 0
    assert np.array_equal(ok, proj64["ok"])
    assert np.array_equal(proj["radii"][ok], proj64["radius"][ok])
    assert np.array_equal(proj["rect"][ok], proj64["rect"][ok])
    np.testing.assert_allclose(proj["means2d"][ok], proj64["mu"][ok], rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(proj["conics"][ok], proj64["conic"][ok], rtol=2e-4, atol=1e-6)
    Wm, amap = ref_np.weights(proj64, g["opac"].astype(np.float64), W0, H0)
    bins = orc.bin_sort(proj, W0, H0)
    gid, pix, w, alphas = orc.blend_pairs(proj, bins, g["opac"], W0, H0, want_alphas=True)
    Wo = np.zeros_like(Wm)
    Wo[pix, gid] = w
    # the set of contributing pairs may differ only where alpha sits within fp32 noise of a threshold
    differ = (Wo > 0) != (Wm > 0)
    assert differ.sum() <= 3, differ.sum()
    np.testing.assert_allclose(Wo[~differ], Wm[~differ], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(alphas, amap, atol=5e-3 if differ.any() else 2e-5)
    if not differ.any():
        F64 = Wm.T @ g["feats"][0].reshape(-1, 8).astype(np.float64)
        F = np.zeros((256, 8), np.float64)
        d = np.zeros(256, np.float64)
        orc.blend_scatter(proj, bins, g["opac"], g["feats"][0], F, d, W0, H0)
        assert rel_row_err(F, F64) < 1e-4
        np.testing.assert_allclose(d, Wm.sum(0), rtol=2e-4, atol=1e-7)


def test_autograd_formulation_equals_direct_accumulation(orc, gold):
    """backproject.py:115-151 literally: render is linear in colours, so with zero colours
    d((render*feats).sum())/d(colors) = W^T feats and d(render.sum())/d(colors0)[:,0] = W^T 1."""
    import ref_np
    g = gold
    proj64 = ref_np.project(g["means"], g["quats"], g["scales"], g["vms"][1], g["K"], W0, H0)
    Wm, _ = ref_np.weights(proj64, g["opac"].astype(np.float64), W0, H0)
    Wt = torch.from_numpy(Wm)
    feats = torch.from_numpy(g["feats"][1].astype(np.float64)).reshape(-1, 8)
    colors_feats = torch.zeros(256, 8, dtype=torch.float64, requires_grad=True)
    colors_feats_0 = torch.zeros(256, 3, dtype=torch.float64, requires_grad=True)
    out = Wt @ colors_feats  # what rasterization() computes for [N,D] colours
    (out * feats).sum().backward()
    (Wt @ colors_feats_0).sum().backward()
    F_auto, d_auto = colors_feats.grad.numpy(), colors_feats_0.grad[:, 0].numpy()
    assert torch.equal(colors_feats_0.grad[:, 0], colors_feats_0.grad[:, 2])  # channel-equal gradients
    proj = orc.project(g["means"], g["quats"], g["scales"], g["vms"][1], g["K"], W0, H0)
    bins = orc.bin_sort(proj, W0, H0)
    F = np.zeros((256, 8), np.float64)
    d = np.zeros(256, np.float64)
    orc.blend_scatter(proj, bins, g["opac"], g["feats"][1], F, d, W0, H0)
    assert rel_row_err(F, F_auto) < 1e-3  # threshold flips (<= 3 pairs) allowed, cf. previous test
    np.testing.assert_allclose(d, d_auto, rtol=2e-3, atol=2e-3)


# ---- closed-form known-answer cases (single camera at the origin looking down +z) ----------------------------
def _cam(W=64, H=48, f=60.0):
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32)
    return np.eye(4, dtype=np.float32), K, W, H


def _run(orc, means, scales, opac, feats=None, quats=None):
    vm, K, W, H = _cam()
    means, scales, opac = np.asarray(means, np.float32), np.asarray(scales, np.float32), np.asarray(opac, np.float32)
    n = means.shape[0]
    quats = np.tile(np.array([2.0, 0, 0, 0], np.float32), (n, 1)) if quats is None else quats  # unnormalised
    proj = orc.project(means, quats, scales, vm, K, W, H)
    bins = orc.bin_sort(proj, W, H)
    gid, pix, w, alphas = orc.blend_pairs(proj, bins, opac, W, H, want_alphas=True)
    return proj, bins, gid, pix, w, alphas


def test_kat_single_isotropic_gaussian(orc):
    s, z, o, f = 0.1, 2.0, 0.8, 60.0
    proj, bins, gid, pix, w, _ = _run(orc, [[0.0, 0.0, z]], [[s, s, s]], [o])
    var = (s * f / z) ** 2 + 0.3  # J Sigma J^T + eps2d, exact for a Gaussian on the optical axis
    assert proj["radii"][0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(proj["conics"][0], [1 / var, 0, 1 / var], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(proj["means2d"][0], [32.0, 24.0], atol=1e-5)
    ys, xs = np.divmod(np.arange(64 * 48), 64)
    r2 = (xs + 0.5 - 32.0) ** 2 + (ys + 0.5 - 24.0) ** 2
    alpha = np.minimum(0.999, o * np.exp(-r2 / (2 * var)))
    x0, y0, x1, y1 = proj["rect"][0]
    inrect = (xs // 16 >= x0) & (xs // 16 < x1) & (ys // 16 >= y0) & (ys // 16 < y1)
    expect = np.where((alpha >= 1 / 255) & inrect, alpha, 0.0)
    got = np.zeros(64 * 48)
    got[pix] = w
    np.testing.assert_allclose(got, expect, rtol=2e-5, atol=1e-7)  # T = 1 everywhere: w = alpha
    assert (got > 0).sum() == len(w) > 50


def test_kat_occlusion_clamp_threshold_and_termination(orc):
    # four big Gaussians stacked on the axis: alpha clamps to 0.999 at the centre pixel(s)
    z = [1.0, 1.5, 2.0, 2.5]
    means = [[0, 0, zz] for zz in z]
    scales = [[1.0, 1.0, 1.0]] * 4
    proj, bins, gid, pix, w, alphas = _run(orc, means, scales, [1.0, 1.0, 1.0, 1.0])
    centre = 24 * 64 + 32
    mine = sorted(zip(gid[pix == centre], w[pix == centre]))
    # front: w = 0.999; second: T = 1e-3 -> T' = 1e-6 <= 1e-4 => terminating Gaussian NOT counted, rest skipped
    assert [g for g, _ in mine] == [0]
    assert abs(mine[0][1] - 0.999) < 1e-6
    assert abs(alphas.reshape(-1)[centre] - 0.999) < 1e-6
    # alpha < 1/255 is skipped: faint Gaussian in front contributes nothing and does not attenuate
    proj, bins, gid, pix, w, _ = _run(orc, [[0, 0, 1.0], [0, 0, 2.0]], [[0.05] * 3, [0.05] * 3], [0.0039, 0.5])
    assert set(gid) == {1}
    vm, K, W, H = _cam()
    var = (0.05 * 60 / 2.0) ** 2 + 0.3
    assert abs(w[pix == centre][0] - 0.5 * math.exp(-0.25 / var)) < 1e-6  # r^2 = 0.5 at the pixel centre


def test_kat_culling_tile_rectangle_and_depth_ties(orc):
    vm, K, W, H = _cam()
    # behind the near plane / off-screen / visible
    proj, *_ = _run(orc, [[0, 0, 0.005], [50, 0, 2.0], [0, 0, 2.0]], [[0.05] * 3] * 3, [0.5] * 3)
    assert list(proj["radii"] > 0) == [False, False, True]
    # a Gaussian is evaluated in every tile of its rectangle and in no other tile
    proj, bins, gid, pix, w, _ = _run(orc, [[0.0, 0.0, 2.0]], [[0.12] * 3], [0.95])
    x0, y0, x1, y1 = proj["rect"][0]
    assert bins["n_isect"] == (x1 - x0) * (y1 - y0)
    tiles = set(zip((pix % 64) // 16, (pix // 64) // 16))
    assert all(x0 <= tx < x1 and y0 <= ty < y1 for tx, ty in tiles)
    # equal depths keep ascending Gaussian index (stable sort): the lower index is composited first
    proj, bins, gid, pix, w, _ = _run(orc, [[0, 0, 2.0], [0, 0, 2.0]], [[0.2] * 3] * 2, [0.6, 0.6])
    centre = 24 * 64 + 32
    sel = pix == centre
    order = dict(zip(gid[sel], w[sel]))
    assert order[0] > order[1] and abs(order[1] / order[0] - (1 - order[0])) < 1e-5
    t = bins["tile_offsets"]
    fl = bins["flatten_ids"]
    for a, b in zip(t[:-1], t[1:]):
        if b - a == 2:
            assert list(fl[a:b]) == [0, 1]


def test_finalize_matches_reference_lines(orc):
    """backproject.py:166-169 in torch vs orc_finalize (fp32 division chain, NaN -> 0)."""
    rng = np.random.default_rng(3)
    F = rng.standard_normal((50, 16))
    d = np.abs(rng.standard_normal(50)) + 0.1
    F[7] = 0
    d[7] = 0
    out = orc.finalize(F, d)
    gf = torch.from_numpy(F).float()
    den = torch.ones(50) * 1e-12 + torch.from_numpy(d).float()
    x = gf / den[..., None]
    x = x / x.norm(dim=-1, keepdim=True)
    x[torch.isnan(x)] = 0
    assert np.abs(out - x.numpy()).max() < 2e-7 and np.all(out[7] == 0)
    assert np.abs(gsbp_amd.finalize_reference(gf, torch.from_numpy(d).float()).numpy() - out).max() < 2e-7


from util import capture_tool  # noqa: E402

CAPTURES = capture_tool().CASES  # (file, config or None = g0.npz, feature channels, encoder outputs): one per kernel family


@pytest.mark.parametrize("fname,cfgname,dim,enc_dim", CAPTURES, ids=[c[0] for c in CAPTURES])
def test_oracle_against_gsplat_capture(orc, fname, cfgname, dim, enc_dim):
    """PINS THE ORACLE when a capture of real gsplat 1.4.0 output exists (tools/capture_gsplat_fixture.py, run by a
    maintainer with CUDA; only the resulting .npz data is committed).  Without the file the oracle stays PARITY UNPINNED
    and this test is skipped.  Threshold rows (pairs exactly at the alpha >= 1/255 or T' <= 1e-4 cut) are expected over
    1e-4 at the rate profiles/r2_sensitivity_C1.json measured for a 2-ulp exp (2-3 rows in 10 000): the bar is on the
    bulk -- 99 % of the rows within 1e-4, at most 0.2 % of the rows (and never more than 1e-2) beyond it."""
    from util import capture_report
    path = os.path.join(os.path.dirname(GOLD), fname)
    if not os.path.exists(path):
        pytest.skip(f"{fname} not captured yet (needs CUDA + gsplat==1.4.0): oracle parity stays unpinned")
    cap = dict(np.load(path))
    if cfgname is None:
        g = dict(np.load(GOLD))
        W, H = W0, H0
    else:
        cfg = syn.CONFIGS[cfgname]
        g = capture_tool().case_inputs(cfgname, dim, enc_dim)
        W, H = cfg.width, cfg.height
    feats = g["feats"] if g.get("encoder") is None else g["feats"] @ g["encoder"]  # backproject_compressed.py:127
    D = feats.shape[-1]
    out, F, d, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W, H,
                                          lambda v: feats[v], D)
    proj = orc.project(g["means"], g["quats"], g["scales"], g["vms"][0], g["K"], W, H)
    rep = capture_report(cap, out, F, d, proj["radii"], proj["means2d"], proj["conics"], proj["depths"])
    print("oracle vs gsplat capture", fname, rep)
    for k in ("F", "d", "out"):
        assert rep[k]["p99"] <= 1e-4, (k, rep[k])
        assert rep[k]["rows_over_1e-4"] <= max(1, int(0.002 * rep[k]["rows"])) and rep[k]["max"] <= 1e-2, (k, rep[k])
    assert rep.get("radii", {}).get("visible_equal", True), rep["radii"]


TOKEN_CAPTURES = capture_tool().TOKEN_CASES


@pytest.mark.parametrize("fname,cfgname,dim,grid", TOKEN_CAPTURES, ids=[c[0] for c in TOKEN_CAPTURES])
def test_oracle_against_gsplat_capture_of_the_dino_loop(orc, fname, cfgname, dim, grid):
    """The same for the dino loop (backproject.py:242-289: nearest-upsampled tokens, .mean() reductions) -- the case the round-6
    token-space kernels are compared with; skipped until a capture exists."""
    from util import capture_report
    path = os.path.join(os.path.dirname(GOLD), fname)
    if not os.path.exists(path):
        pytest.skip(f"{fname} not captured yet (needs CUDA + gsplat==1.4.0): oracle parity stays unpinned")
    cap = dict(np.load(path))
    cfg = syn.CONFIGS[cfgname]
    g = capture_tool().token_case_inputs(cfgname, dim, grid)
    ups = [torch.nn.functional.interpolate(torch.from_numpy(f).permute(2, 0, 1)[None], size=(cfg.height, cfg.width),
                                           mode="nearest")[0].permute(1, 2, 0).contiguous().numpy() for f in g["feats"]]
    out, F, d, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], cfg.width, cfg.height,
                                          lambda v: ups[v], dim, reduction="mean")
    rep = capture_report(cap, out, F, d)
    print("oracle vs gsplat capture", fname, rep)
    for k in ("F", "d", "out"):
        assert rep[k]["p99"] <= 1e-4, (k, rep[k])
        assert rep[k]["rows_over_1e-4"] <= max(1, int(0.002 * rep[k]["rows"])) and rep[k]["max"] <= 1e-2, (k, rep[k])


def test_capture_script_stays_in_sync_with_its_consumers(orc, gold, tmp_path, monkeypatch):
    """tools/capture_gsplat_fixture.py cannot run here (it needs CUDA + gsplat 1.4.0); what CAN rot unnoticed is the
    agreement between the keys it saves and the keys the two consumer tests read (capture_report).  Run its capture()
    against a stand-in `gsplat` module whose rasterization() is the oracle in the reference's autograd formulation (a dense
    weight matrix times the differentiable colour table) and feed the file it writes to capture_report: every quantity
    must be found and -- the stand-in being the oracle -- agree."""
    import importlib.util
    import sys
    import types
    from util import capture_report
    g = gold

    def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, **kw):
        m, q, s, o = (t.detach().numpy() for t in (means, quats, scales, opacities))
        proj = orc.project(m, q, s, viewmats[0].numpy(), Ks[0].numpy(), width, height)
        bins = orc.bin_sort(proj, width, height)
        gid, pix, w, alphas = orc.blend_pairs(proj, bins, o, width, height, want_alphas=True)
        Wm = torch.zeros(width * height, m.shape[0], dtype=torch.float64)
        Wm[torch.from_numpy(pix.astype(np.int64)), torch.from_numpy(gid.astype(np.int64))] = torch.from_numpy(w.astype(np.float64))
        out = (Wm @ colors.double()).to(colors.dtype).reshape(1, height, width, colors.shape[1])
        meta = {k: torch.from_numpy(np.ascontiguousarray(proj[k]))[None] for k in ("means2d", "radii", "conics", "depths")}
        meta.update(isect_ids=torch.from_numpy(bins["isect_ids"]), flatten_ids=torch.from_numpy(bins["flatten_ids"]),
                    tile_size=16, tile_width=-(-width // 16), tile_height=-(-height // 16), width=width, height=height)
        return out, torch.from_numpy(alphas)[None, ..., None], meta

    stub = types.ModuleType("gsplat")
    stub.rasterization, stub.__version__ = rasterization, "stand-in (oracle)"
    monkeypatch.setitem(sys.modules, "gsplat", stub)
    monkeypatch.setenv("GWBP_CAPTURE_DEVICE", "cpu")
    spec = importlib.util.spec_from_file_location(
        "capture_gsplat_fixture", os.path.join(os.path.dirname(os.path.dirname(GOLD)), "..", "tools", "capture_gsplat_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = str(tmp_path / "gsplat_g0.npz")
    mod.capture({k: g[k] for k in ("means", "quats", "scales", "opac", "K", "vms", "feats")}, path)
    cap = dict(np.load(path))
    out, F, d, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W0, H0,
                                          lambda v: g["feats"][v], 8)
    proj = orc.project(g["means"], g["quats"], g["scales"], g["vms"][0], g["K"], W0, H0)
    rep = capture_report(cap, out, F, d, proj["radii"], proj["means2d"], proj["conics"], proj["depths"])
    for k in ("F", "d", "out"):
        assert rep[k]["rows"] == 256 and rep[k]["max"] <= 1e-5, (k, rep[k])
    assert rep["radii"] == {"visible_equal": True, "n_diff": 0}
    for k in ("means2d", "conics", "depths"):
        assert rep[k]["max_rel"] == 0.0, (k, rep[k])
    for k in ("v0_alphas", "v0_isect_ids", "v0_flatten_ids", "F_views", "d_views", "gsplat_version"):
        assert k in cap, k
    # the compressed case (feats @ encoder, then D = 16) and the table the consumers are parametrised over
    assert [c[0] for c in mod.CASES] == ["gsplat_g0.npz", "gsplat_t1.npz", "gsplat_c1.npz", "gsplat_t1_d64enc16.npz",
                                         "gsplat_t1_d128.npz", "gsplat_t1_d256.npz"]
    enc = (np.random.default_rng(3).standard_normal((8, 4)) / 8 ** 0.5).astype(np.float32)
    path2 = str(tmp_path / "gsplat_enc.npz")
    mod.capture({**{k: g[k] for k in ("means", "quats", "scales", "opac", "K", "vms", "feats")}, "encoder": enc}, path2,
                per_view=False)
    cap2 = dict(np.load(path2))
    fe = g["feats"] @ enc
    out2, F2, d2, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W0, H0,
                                             lambda v: fe[v], 4)
    rep2 = capture_report(cap2, out2, F2, d2)
    assert cap2["F"].shape == (256, 4) and "F_views" not in cap2 and all(rep2[k]["max"] <= 1e-5 for k in ("F", "d", "out"))
    # the dino loop (round 6: the token-space kernels' case): an upsampled token map and .mean() reductions through capture()
    assert [c[0] for c in mod.TOKEN_CASES] == ["gsplat_t1_tokens8x12_d256.npz"]
    tok = np.random.default_rng(4).standard_normal((g["vms"].shape[0], 3, 4, 8)).astype(np.float32)
    path3 = str(tmp_path / "gsplat_tok.npz")
    mod.capture({**{k: g[k] for k in ("means", "quats", "scales", "opac", "K", "vms")}, "feats": tok, "upsample": "nearest",
                 "reduction": "mean"}, path3, per_view=False)
    cap3 = dict(np.load(path3))
    ups = [torch.nn.functional.interpolate(torch.from_numpy(f).permute(2, 0, 1)[None], size=(H0, W0), mode="nearest")[0]
           .permute(1, 2, 0).contiguous().numpy() for f in tok]
    out3, F3, d3, _ = orc.backproject_oracle(g["means"], g["quats"], g["scales"], g["opac"], g["vms"], g["K"], W0, H0,
                                             lambda v: ups[v], 8, reduction="mean")
    rep3 = capture_report(cap3, out3, F3, d3)
    assert all(rep3[k]["max"] <= 1e-5 for k in ("F", "d", "out")), rep3
    tk = mod.token_case_inputs("T1", 256, (8, 12))
    assert tk["feats"].shape == (syn.CONFIGS["T1"].n_views, 8, 12, 256) and tk["reduction"] == "mean"
    t1 = mod.case_inputs("T1", 64, 16)
    assert t1["feats"].shape[-1] == 64 and t1["encoder"].shape == (64, 16) and t1["means"].shape == (syn.CONFIGS["T1"].n_gaussians, 3)
