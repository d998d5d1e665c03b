"""Randomised sweep of small scenes through the whole HIP path against the CPU oracle: image sizes that are not
multiples of the tile, odd channel counts on every scatter kernel (D <= 64, D % 128 == 0, generic), Gaussians from
sub-pixel to screen-filling, views that cull most of the scene, channel-major and padded feature-map strides."""
import math

import numpy as np
import pytest
import torch

from util import rel_row_err

import gsbp_amd
from gsbp_amd import synthetic as syn

pytestmark = pytest.mark.gpu

CASES = [
    # (seed, N, W, H, D, log_scale0, layout)
    (1, 300, 17, 9, 5, 0.20, "hwc"),        # smaller than two tiles, huge Gaussians (every tile touched)
    (2, 1000, 33, 47, 64, 0.05, "chw"),     # D = 64 boundary of the small path, channel-major map
    (3, 2000, 130, 70, 65, 0.03, "hwc"),    # generic kernel (65 channels)
    (4, 1500, 96, 80, 128, 0.02, "chw"),    # one 128-chunk, channel-major: element-wise staging
    (5, 800, 64, 64, 256, 0.10, "padded"),  # two chunks, padded pixel pitch
    (6, 5000, 250, 100, 1, 0.004, "hwc"),   # one channel, sub-pixel Gaussians
    (7, 64, 16, 16, 384, 0.50, "hwc"),      # one tile, three chunks, screen-filling Gaussians (128+ entries per record)
    (8, 3000, 300, 20, 40, 0.03, "padded"), # wide strip
    (9, 1200, 50, 200, 130, 0.05, "chw"),   # tall strip, generic kernel with strides
    (10, 1, 40, 40, 7, 0.30, "hwc"),        # a single Gaussian
    # the 256-channel kernel (set_narrow_scatter(False): half-tile lists, carry rows), D % 256 == 0, channel-contiguous maps
    (11, 2500, 131, 77, 256, 0.03, "wide"),        # edge tiles on both axes
    (12, 64, 16, 16, 512, 0.50, "wide"),           # one tile, two chunks, screen-filling Gaussians (128-entry half-tile visits)
    (13, 6000, 250, 100, 256, 0.004, "wide"),      # sub-pixel Gaussians: one- and two-pixel records, few spanning both halves
    (14, 17, 48, 48, 256, 0.15, "wide_padded"),    # a handful of records per tile, padded pixel pitch
]


def _scene(seed, n, log_scale0):
    g = torch.Generator().manual_seed(1000 + seed)
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    scales = torch.exp(math.log(log_scale0) + 0.7 * torch.randn(n, 3, generator=g))
    quats = torch.randn(n, 4, generator=g)
    opac = torch.sigmoid(2.0 * torch.randn(n, generator=g))
    return means, quats, scales, opac


def _camera(seed, W, H):
    g = torch.Generator().manual_seed(2000 + seed)
    th = float(torch.rand(1, generator=g)) * 2 * math.pi
    el = math.radians(10 + 50 * float(torch.rand(1, generator=g)))
    r = 1.5 + 3.0 * float(torch.rand(1, generator=g))  # some cameras sit inside the cloud: near-plane culling
    c = torch.tensor([r * math.cos(th) * math.cos(el), r * math.sin(th) * math.cos(el), r * math.sin(el)])
    fwd = -c / c.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd])
    vm = torch.eye(4)
    vm[:3, :3] = R
    vm[:3, 3] = -R @ c
    f = (0.6 + 1.2 * float(torch.rand(1, generator=g))) * W
    K = torch.tensor([[f, 0, W / 2 + 3.3 * float(torch.rand(1, generator=g))], [0, 0.9 * f, H / 2 - 1.7], [0, 0, 1.0]])
    return vm, K


def _layout(feats, layout, dev):
    f = feats.to(dev)
    if layout == "chw":
        return f.permute(2, 0, 1).contiguous().permute(1, 2, 0)
    if layout == "padded":
        H, W, D = f.shape
        buf = torch.zeros(H, W + 3, D + 4, device=dev)
        buf[:, :W, :D] = f
        return buf[:, :W, :D]
    return f


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}_N{c[1]}_{c[2]}x{c[3]}_D{c[4]}_{c[6]}" for c in CASES])
def test_random_scene_matches_oracle(case, orc, dev):
    seed, n, W, H, D, s0, layout = case
    means, quats, scales, opac = _scene(seed, n, s0)
    wide = layout.startswith("wide")
    eng = gsbp_amd.Engine(n, W, H, device=dev)
    eng.set_narrow_scatter(not wide)
    layout = {"wide": "hwc", "wide_padded": "padded"}.get(layout, layout)
    F = torch.zeros(n, D, device=dev)
    d = torch.zeros(n, device=dev)
    Fr = np.zeros((n, D), np.float64)
    dr = np.zeros(n, np.float64)
    g_dev = [t.to(dev) for t in (means, quats, scales, opac)]
    g_np = [t.numpy() for t in (means, quats, scales, opac)]
    total_pairs = 0
    for v in range(3):
        vm, K = _camera(10 * seed + v, W, H)
        feats = torch.randn(H, W, D, generator=torch.Generator().manual_seed(3000 + 10 * seed + v))
        if D < 4:  # a row of one or two signed terms can cancel to ~0: the row-relative metric needs positive data there
            feats = feats.abs()
        eng.backproject_view(eng.view(vm, K, W, H), *g_dev, _layout(feats, layout, dev), F, d)
        st = eng.stats()
        info = orc.backproject_view(*g_np, vm.numpy(), K.numpy(), W, H, feats.numpy(), Fr, dr)
        assert st["overflow"] == 0 and st["blend_kind"] == (1 if wide else 0)
        assert (st["n_pairs"], st["n_isect"], st["n_visible"]) == (info["n_pairs"], info["n_isect"], info["n_vis"])
        total_pairs += info["n_pairs"]
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    out = eng.finalize(F, d).cpu().numpy()
    ref = orc.finalize(np.ascontiguousarray(Fr), dr)
    ok = dr > 1e-6 * max(dr.max(), 1e-30)
    if ok.any():
        assert rel_row_err(out[ok], ref[ok]) <= 1e-4
    assert np.array_equal(out[dr == 0], np.zeros_like(out[dr == 0]))


@pytest.mark.parametrize("wgs", [8, 24, 1000])
def test_scatter_grid_size_does_not_change_the_result(wgs, orc, dev):
    """caps.scatter_workgroups: any persistent grid (fewer workgroups than XCD classes x tiles, more than CUs) must
    drain the per-class queues completely and re-arm them (two scatters on one weight store)."""
    seed, n, W, H, D, s0 = 42, 3000, 200, 150, 128, 0.03
    means, quats, scales, opac = _scene(seed, n, s0)
    vm, K = _camera(seed, W, H)
    feats = torch.randn(H, W, D, generator=torch.Generator().manual_seed(7))
    eng = gsbp_amd.Engine(n, W, H, device=dev, scatter_workgroups=wgs)
    view = eng.view(vm, K, W, H)
    g_dev = [t.to(dev) for t in (means, quats, scales, opac)]
    eng.project(view, *g_dev)
    eng.bin_sort(view)
    eng.blend_weights(view)
    F = torch.zeros(n, D, device=dev)
    d = torch.zeros(n, device=dev)
    eng.scatter(view, feats.to(dev), F, d)
    eng.scatter(view, feats.to(dev), F, d)  # same store again: the queues must have re-armed themselves
    Fr = np.zeros((n, D), np.float64)
    dr = np.zeros(n, np.float64)
    orc.backproject_view(*[t.numpy() for t in (means, quats, scales, opac)], vm.numpy(), K.numpy(), W, H,
                         feats.numpy(), Fr, dr)
    assert rel_row_err(F.cpu().numpy(), 2.0 * Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], 2.0 * dr[:, None]) <= 1e-4


WIDE_CASES = [
    # (seed, N, W, H, D, log_scale0): the 256-channel scatter kernel (half-tile slabs, carried partial sums)
    (21, 2000, 130, 70, 256, 0.03),    # one chunk; records in one or both halves of a tile
    (22, 64, 16, 16, 512, 0.50),       # one tile, screen-filling Gaussians: every record spans both halves, 128 entries per half
    (23, 30000, 48, 48, 256, 0.02),    # > 1024 records per tile: the carry rows run out, the rest is flushed per half
    (24, 1500, 96, 80, 768, 0.02),     # three chunks
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=[f"seed{c[0]}_N{c[1]}_{c[2]}x{c[3]}_D{c[4]}" for c in WIDE_CASES])
def test_wide_scatter_kernel_matches_oracle_and_narrow(case, orc, dev):
    seed, n, W, H, D, s0 = case
    means, quats, scales, opac = _scene(seed, n, s0)
    if seed == 23:  # faint Gaussians: nothing terminates, every tile collects thousands of records
        opac = torch.full_like(opac, 0.03)
    g_dev = [t.to(dev) for t in (means, quats, scales, opac)]
    g_np = [t.numpy() for t in (means, quats, scales, opac)]
    vm, K = _camera(seed, W, H)
    feats = torch.randn(H, W, D, generator=torch.Generator().manual_seed(seed))
    res = {}
    for name, narrow in (("wide", False), ("narrow", True)):
        eng = gsbp_amd.Engine(n, W, H, device=dev, pair_cap=1 << 25, isect_cap=1 << 22)
        eng.set_narrow_scatter(narrow)
        view = eng.view(vm, K, W, H)
        F = torch.zeros(n, D, device=dev)
        d = torch.zeros(n, device=dev)
        eng.backproject_view(view, *g_dev, feats.to(dev), F, d)
        eng.scatter(view, feats.to(dev), F, d)  # the same store again (queues re-armed, carry rows reused)
        assert eng.stats()["overflow"] == 0
        res[name] = (F.cpu().numpy(), d.cpu().numpy(), eng.stats())
    Fr = np.zeros((n, D), np.float64)
    dr = np.zeros(n, np.float64)
    info = orc.backproject_view(*g_np, vm.numpy(), K.numpy(), W, H, feats.numpy(), Fr, dr)
    assert res["wide"][2]["n_pairs"] == info["n_pairs"] == res["narrow"][2]["n_pairs"]
    if seed == 23:
        assert res["wide"][2]["n_headers"] > 2 * 1024 * 9  # 9 tiles, each well beyond the 1024 carry rows
    for name in ("wide", "narrow"):
        assert rel_row_err(res[name][0], 2.0 * Fr) <= 1e-4, name
        assert rel_row_err(res[name][1][:, None], 2.0 * dr[:, None]) <= 1e-4, name


def test_wide_scatter_kernel_with_nearest_upsampled_map(orc, dev):
    """The 256-channel kernel reading a low-resolution map through the index maps (gwbp_scatter_upsampled, D = 256)."""
    seed, n, W, H, D, s0 = 31, 2500, 150, 90, 256, 0.03
    means, quats, scales, opac = _scene(seed, n, s0)
    vm, K = _camera(seed, W, H)
    low = torch.randn(11, 19, D, generator=torch.Generator().manual_seed(seed))
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), mode="nearest")[0].permute(1, 2, 0)
    eng = gsbp_amd.Engine(n, W, H, device=dev)
    eng.set_narrow_scatter(False)
    view = eng.view(vm, K, W, H)
    eng.project(view, *[t.to(dev) for t in (means, quats, scales, opac)])
    eng.bin_sort(view)
    eng.blend_weights(view)
    F = torch.zeros(n, D, device=dev)
    d = torch.zeros(n, device=dev)
    eng.scatter(view, low.to(dev), F, d, upsample="nearest")
    d2 = torch.zeros(n, device=dev)
    eng.accumulate_d(view, d2)  # the denominators alone, from the blend's weight sums
    Fr = np.zeros((n, D), np.float64)
    dr = np.zeros(n, np.float64)
    orc.backproject_view(*[t.numpy() for t in (means, quats, scales, opac)], vm.numpy(), K.numpy(), W, H,
                         np.ascontiguousarray(up.numpy()), Fr, dr)
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4
    assert torch.allclose(d, d2, rtol=1e-5, atol=0)
    eng.set_narrow_scatter(True)
    with pytest.raises(gsbp_amd.GwbpError):
        eng.accumulate_d(view, d2)  # a narrow blend leaves no weight sums


@pytest.mark.parametrize("D", [512, 48, 130], ids=["D512_fast_kernel", "D48_small_kernel", "D130_generic_kernel"])
def test_scatter_bilinear_upsampled_lowres_map(orc, dev, D):
    """lseg variant (backproject.py:108-113): the normalised low-resolution map [h,w,D] is upsampled bilinearly
    (align_corners=False) while the slabs are staged; the oracle gets F.interpolate's materialised map."""
    seed, n, W, H, s0 = 41, 2500, 150, 90, 0.03
    means, quats, scales, opac = _scene(seed, n, s0)
    vm, K = _camera(seed, W, H)
    low = torch.nn.functional.normalize(torch.randn(23, 31, D, generator=torch.Generator().manual_seed(seed)), dim=2)
    up = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                         align_corners=False)[0].permute(1, 2, 0)
    eng = gsbp_amd.Engine(n, W, H, device=dev)
    view = eng.view(vm, K, W, H)
    F = torch.zeros(n, D, device=dev)
    d = torch.zeros(n, device=dev)
    eng.project(view, *[t.to(dev) for t in (means, quats, scales, opac)])
    eng.bin_sort(view)
    eng.blend_weights(view)
    eng.scatter(view, low.to(dev), F, d, upsample="bilinear")
    Fr = np.zeros((n, D), np.float64)
    dr = np.zeros(n, np.float64)
    orc.backproject_view(*[t.numpy() for t in (means, quats, scales, opac)], vm.numpy(), K.numpy(), W, H,
                         np.ascontiguousarray(up.numpy()), Fr, dr)
    assert rel_row_err(F.cpu().numpy(), Fr) <= 1e-4
    assert rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) <= 1e-4
