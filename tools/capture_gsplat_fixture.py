#!/usr/bin/env python3
"""PATH TO PINNED PARITY: capture what the REAL reference arithmetic (gsplat 1.4.0, CUDA) produces on this repository's
golden inputs, so that the oracle and the HIP kernels can be compared with it.

Cannot run in this project's containers (gsplat is CUDA-only, not vendored, no network).  A maintainer with an NVIDIA GPU
runs, once:

    pip install gsplat==1.4.0 torch numpy
    python tools/capture_gsplat_fixture.py            # reads tests/golden/g0.npz and this package's seeded generators,
                                                      # writes tests/golden/gsplat_*.npz, one file per entry of CASES below

One run pins every kernel family: g0 / T1 (D = 8 / 24: the general scatter kernel), C1 IN FULL (BASELINE config 1: 10 000
Gaussians, 4 views 400 x 300, D = 32: the fused blend + scatter kernels), T1 with 64-channel maps through a 64 -> 16 encoder
(backproject_compressed.py:127: encoder kernel + fused small-D kernel), T1 at D = 128 (k_scatter_full) and at D = 256
(k_scatter_wide); round 6: the DINO loop (backproject.py:242-289: an 8 x 12 x 256 token map nearest-upsampled to the view,
.mean() reductions) on T1 -- the token-space kernels (k_blend<kToken> + k_token_apply), TOKEN_CASES below.

and commits the resulting .npz DATA files (inputs are already committed; nothing of gsplat's or the reference's source
travels).  tests/test_oracle.py::test_oracle_against_gsplat_capture (CPU) and tests/test_gpu_parity.py::
test_hip_against_gsplat_capture (GPU) consume the files when present and are skipped otherwise.

What is captured, per view, mirrors the reference's per-view body (backproject.py:115-151) literally:
  * rasterization(means, quats, scales, opacities, zeros[N, D], viewmat[None], K[None], width=W, height=H)
    -> (render * feats).sum().backward() -> colors.grad                      = F_v   [N, D]
  * rasterization(..., zeros[N, 3], ...) -> render.sum().backward() -> grad[:, 0] = d_v   [N]
  * the forward meta of view 0: means2d, radii, conics, depths, isect_ids, flatten_ids (+ tile geometry), render alphas
and the finalised field  normalize((sum F_v) / (1e-12 + sum d_v)), NaN -> 0  (backproject.py:62-63,166-169).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# (file, seeded config or None = tests/golden/g0.npz, feature channels or None = the config's, encoder outputs or None)
CASES = [
    ("gsplat_g0.npz", None, None, None),
    ("gsplat_t1.npz", "T1", None, None),
    ("gsplat_c1.npz", "C1", None, None),
    ("gsplat_t1_d64enc16.npz", "T1", 64, 16),
    ("gsplat_t1_d128.npz", "T1", 128, None),
    ("gsplat_t1_d256.npz", "T1", 256, None),
]
# the dino variant: (file, seeded config, token channels, (token rows, token columns)); the capture upsamples the tokens with
# F.interpolate(mode="nearest") and reduces with .mean() exactly like backproject.py:244-248,263,283
TOKEN_CASES = [
    ("gsplat_t1_tokens8x12_d256.npz", "T1", 256, (8, 12)),
]
ENCODER_SEED = 7
TOKEN_SEED0 = 500


def case_inputs(cfgname, dim=None, enc_dim=None):
    """numpy inputs of a capture case (shared with the consumer tests): the seeded scene, cameras and feature maps of this
    repository's generators (gsbp_amd.synthetic: torch only, no HIP library needed), plus -- compressed variant -- a seeded
    [dim, enc_dim] encoder N(0, 1) / sqrt(dim) like synthetic.make_encoder."""
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from gsbp_amd import synthetic as syn
    cfg = syn.CONFIGS[cfgname]
    means, quats, scales, opac = [x.numpy() for x in syn.activate(syn.make_scene(cfg))]
    inp = dict(means=means, quats=quats, scales=scales, opac=opac, K=syn.intrinsics(cfg).numpy(),
               vms=syn.make_cameras(cfg).numpy(),
               feats=torch.stack([syn.make_feature_map(cfg, v, dim=dim) for v in range(cfg.n_views)]).numpy())
    if enc_dim:
        g = torch.Generator(device="cpu").manual_seed(ENCODER_SEED)
        inp["encoder"] = (torch.randn(inp["feats"].shape[-1], enc_dim, generator=g) / inp["feats"].shape[-1] ** 0.5).numpy()
    return inp


def token_case_inputs(cfgname, dim, grid):
    """numpy inputs of a TOKEN_CASES entry: the seeded scene and cameras of `cfgname` and one N(0, 1) token map [h, w, dim] per view
    (dino's patch tokens are not normalised, backproject.py:242-243)."""
    import torch
    inp = case_inputs(cfgname, 1)
    V = inp["vms"].shape[0]
    inp["feats"] = np.stack([torch.randn(grid[0], grid[1], dim, generator=torch.Generator().manual_seed(TOKEN_SEED0 + v)).numpy()
                             for v in range(V)])
    inp["upsample"], inp["reduction"] = "nearest", "mean"
    return inp


def capture(inp, out_path, per_view=True):
    import torch
    from gsplat import rasterization  # the reference's import (backproject.py:7)
    import gsplat
    # (GWBP_CAPTURE_DEVICE=cpu exists for tests/test_oracle.py::test_capture_script_stays_in_sync_with_its_consumers, which runs
    # this function against a stand-in `gsplat` module; a real capture runs on CUDA)
    dev = torch.device(os.environ.get("GWBP_CAPTURE_DEVICE", "cuda"))
    t = {k: torch.from_numpy(np.asarray(inp[k])).to(dev) for k in ("means", "quats", "scales", "opac", "K", "vms", "feats")}
    if inp.get("encoder") is not None:  # backproject_compressed.py:127: feats @ encoder, then the D = 16 loop
        t["feats"] = t["feats"] @ torch.from_numpy(np.asarray(inp["encoder"])).to(dev)
    mean = inp.get("reduction") == "mean"  # dino: .mean() instead of .sum() (backproject.py:263,283)
    if inp.get("upsample") == "nearest":   # dino: the tokens are upsampled to the view first (backproject.py:244-248)
        Wv, Hv = int(2 * float(inp["K"][0][2])), int(2 * float(inp["K"][1][2]))  # backproject.py:85-86
        t["feats"] = torch.nn.functional.interpolate(t["feats"].permute(0, 3, 1, 2), size=(Hv, Wv), mode="nearest").permute(0, 2, 3, 1)
    N, (V, H, W, D) = t["means"].shape[0], t["feats"].shape
    F = torch.zeros(N, D, device=dev)
    d = torch.zeros(N, device=dev)
    meta0, alphas0 = None, None
    Fv, dv = [], []
    for v in range(V):
        colors = torch.zeros(N, D, device=dev, requires_grad=True)
        out, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opac"], colors, t["vms"][v][None],
                                          t["K"][None], width=W, height=H)
        prod = out[0] * t["feats"][v]
        (prod.mean() if mean else prod.sum()).backward()
        Fv.append(colors.grad.detach().clone())
        c3 = torch.zeros(N, 3, device=dev, requires_grad=True)
        out3, _, _ = rasterization(t["means"], t["quats"], t["scales"], t["opac"], c3, t["vms"][v][None], t["K"][None],
                                   width=W, height=H)
        (out3.mean() if mean else out3.sum()).backward()
        dv.append(c3.grad[:, 0].detach().clone())
        F += Fv[-1]
        d += dv[-1]
        if v == 0:
            meta0, alphas0 = meta, alphas
    x = F / (1e-12 + d)[:, None]
    x = x / x.norm(dim=-1, keepdim=True)
    x[torch.isnan(x)] = 0

    def m(key):
        val = meta0.get(key)
        return None if val is None else val.detach().cpu().numpy()

    save = dict(gsplat_version=np.array(gsplat.__version__), torch_version=np.array(torch.__version__),
                device=np.array(torch.cuda.get_device_name(0) if dev.type == "cuda" else str(dev)),
                F=F.cpu().numpy(), d=d.cpu().numpy(), out=x.cpu().numpy(), d_views=torch.stack(dv).cpu().numpy(),
                v0_alphas=alphas0[0, ..., 0].detach().cpu().numpy())
    if per_view:  # (left out for the wide maps: [V, N, 256] is most of the file)
        save["F_views"] = torch.stack(Fv).cpu().numpy()
    # packed=True (gsplat's default) returns per-visible-Gaussian arrays + gaussian_ids; both layouts are stored as given
    for key in ("means2d", "radii", "conics", "depths", "gaussian_ids", "camera_ids", "isect_ids", "flatten_ids",
                "isect_offsets", "tiles_per_gauss"):
        val = m(key)
        if val is not None:
            save["v0_" + key] = val
    for key in ("tile_size", "tile_width", "tile_height", "width", "height"):
        if key in meta0:
            save["v0_" + key] = np.array(int(meta0[key]))
    np.savez_compressed(out_path, **save)
    print("wrote", out_path, {k: getattr(v, "shape", v) for k, v in save.items()})


def main():
    for fname, cfgname, dim, enc_dim in CASES:
        if cfgname is None:
            inp = dict(np.load(os.path.join(GOLD, "g0.npz")))
        else:
            try:
                inp = case_inputs(cfgname, dim, enc_dim)
            except ImportError as e:  # (this package's generators need torch only)
                print(fname, "skipped:", e)
                continue
        capture(inp, os.path.join(GOLD, fname), per_view=(dim or 0) < 128)
    for fname, cfgname, dim, grid in TOKEN_CASES:
        capture(token_case_inputs(cfgname, dim, grid), os.path.join(GOLD, fname), per_view=False)


if __name__ == "__main__":
    main()
