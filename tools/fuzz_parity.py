"""Randomised parity fuzz (GPU + oracle): many small random scenes, image sizes, channel counts, map layouts and both
D % 256 == 0 scatter kernels against the CPU oracle.  Not part of the test suite (minutes of run time).
  python tools/fuzz_parity.py [n_cases] [seed0] [tokens]     ("tokens": every case through the token-space kernels)"""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gsbp_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from util import rel_row_err  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only_tokens = len(sys.argv) > 3 and sys.argv[3] == "tokens"  # every case a nearest-upsampled coarse map with D % 4 == 0
dev = torch.device("cuda:0")
orc.build()
DIMS = [1, 3, 4, 7, 8, 12, 16, 17, 32, 40, 64, 65, 100, 128, 130, 256, 384, 512, 768, 1024]
bad = 0
n_fused = 0
n_enc = 0
n_render = 0
n_px = 0
n_tok = 0
n_split = 0
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(1, 4000)) if rng.random() < 0.9 else int(rng.integers(4000, 30000))  # (depth-sort variants)
    W, H = int(rng.integers(1, 260)), int(rng.integers(1, 200))
    D = int(rng.choice(DIMS))
    if only_tokens:
        D = int(rng.choice([256, 512, 768, 1024, 1280, 1536, 384, 64, 4 * int(rng.integers(16, 520))]))
        W, H = int(rng.integers(16, 400)), int(rng.integers(16, 300))
    s0 = float(10 ** rng.uniform(-2.6, -0.3))
    g = torch.Generator().manual_seed(seed0 + case)
    means = (torch.rand(n, 3, generator=g) * 2 - 1) * float(rng.uniform(0.2, 1.5))
    scales = torch.exp(math.log(s0) + 0.8 * torch.randn(n, 3, generator=g))
    quats = torch.randn(n, 4, generator=g)
    opac = torch.sigmoid(float(rng.uniform(0.5, 3.0)) * torch.randn(n, generator=g) + float(rng.uniform(-3, 2)))
    th, el, r = float(rng.uniform(0, 2 * math.pi)), math.radians(float(rng.uniform(5, 70))), float(rng.uniform(1.2, 5))
    c = torch.tensor([r * math.cos(th) * math.cos(el), r * math.sin(th) * math.cos(el), r * math.sin(el)])
    fwd = -c / c.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
    right = right / right.norm()
    R = torch.stack([right, torch.linalg.cross(fwd, right), fwd])
    vm = torch.eye(4)
    vm[:3, :3], vm[:3, 3] = R, -R @ c
    f = float(rng.uniform(0.5, 2.0)) * max(W, H)
    K = torch.tensor([[f, 0, W / 2 + float(rng.uniform(-3, 3))], [0, f * float(rng.uniform(0.8, 1.2)), H / 2], [0, 0, 1.0]])
    up = rng.choice([None, None, "nearest", "bilinear"])
    if only_tokens:
        up = "nearest"
    if up is not None:  # a low-resolution map; the oracle gets F.interpolate's materialised version
        lh, lw = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if up == "nearest" and D % 4 == 0 and D >= 64 and rng.random() < 0.7:  # round 6: coarse enough for the token-space path
            lh, lw = int(rng.integers(1, max(2, H // 16 + 1))), int(rng.integers(1, max(2, W // 16 + 1)))
        low = torch.randn(lh, lw, D, generator=g)
        if D < 4:
            low = low.abs()
        kw = dict(mode="nearest") if up == "nearest" else dict(mode="bilinear", align_corners=False)
        feats = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), **kw)[0].permute(1, 2, 0).contiguous()
    else:
        feats = torch.randn(H, W, D, generator=g)
        if D < 4:
            feats = feats.abs()
    layout = rng.choice(["hwc", "chw", "pad"])
    # the compressed variant in one kernel (round 5): a K-channel map and a [K, D] encoder, D <= 16; the oracle gets feats_in @ enc
    enc = None
    if up is None and D <= 16 and rng.random() < 0.4:
        Kin = int(rng.choice([16, 32, 48, 64, 128, 512]))
        feats_in = torch.randn(H, W, Kin, generator=g)
        enc = torch.randn(Kin, D, generator=g) / Kin ** 0.5
        if D < 4:  # (as for plain maps: rows of one to three signed values cancel and the relative row error means nothing)
            feats_in, enc = feats_in.abs(), enc.abs()
        feats = feats_in @ enc
        layout = "pad" if layout == "chw" else layout  # (channel-contiguous pixels required)
    fd = (low if up is not None else (feats if enc is None else feats_in)).to(dev)
    if layout == "chw":
        fd = fd.permute(2, 0, 1).contiguous().permute(1, 2, 0)
    elif layout == "pad":
        Dm = fd.shape[2]
        buf = torch.zeros(fd.shape[0], fd.shape[1] + 2, Dm + 4, device=dev)
        buf[:, :fd.shape[1], :Dm] = fd
        fd = buf[:, :fd.shape[1], :Dm]
    wide = bool(rng.integers(0, 2))
    tight = bool(rng.integers(0, 2))
    eng = gsbp_amd.Engine(n, W, H, device=dev, tight_binning=tight, isect_cap=1 << 21, pair_cap=1 << 24)
    eng.set_narrow_scatter(not wide)
    view = eng.view(vm, K, W, H)
    fused = enc is None and up is None and bool(rng.integers(0, 2)) and gsbp_amd.Engine.can_blend_scatter(fd)
    # round 6: nearest-upsampled maps whose texels cover a tile go through token space (gwbp_blend_tokens + gwbp_scatter_tokens);
    # the encoder-fused kernel in its producer / consumer form (GWBP_FLAG_SPLIT_ENCODER) on every second encoder case
    tok = up == "nearest" and gsbp_amd.Engine.can_scatter_tokens(fd, H, W) and rng.random() < 0.8
    split = enc is not None and bool(rng.integers(0, 2))
    eng.set_split_encoder(split)
    for attempt in range(4):  # a capacity overflow invalidates the view: grow the workspace and run it again
        F = torch.zeros(n, D, device=dev)
        d = torch.zeros(n, device=dev)
        if enc is not None:  # encoder + blend + scatter in one kernel (gwbp_blend_scatter_encoded)
            eng.project(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
            eng.bin_sort(view)
            eng.blend_scatter_encoded(view, fd, enc.to(dev), F, d)
        elif fused:  # D <= 16, unit channel stride: blend + scatter in one kernel (gwbp_blend_scatter)
            eng.project(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
            eng.bin_sort(view)
            eng.blend_scatter(view, fd, F, d)
        elif tok:
            eng.project(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
            eng.bin_sort(view)
            eng.blend_tokens(view, fd.shape[0], fd.shape[1])
            eng.scatter_tokens(view, fd, F, d)
        elif up is None:
            eng.backproject_view(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev), fd, F, d)
        else:
            eng.project(view, means.to(dev), quats.to(dev), scales.to(dev), opac.to(dev))
            eng.bin_sort(view)
            eng.blend_weights(view)
            eng.scatter(view, fd, F, d, upsample=str(up))
        if not eng.stats()["overflow"]:
            break
        eng.grow(eng.stats())
        eng.set_narrow_scatter(not wide)
        eng.set_split_encoder(split)
    n_fused += int(fused)
    n_enc += int(enc is not None)
    n_tok += int(tok)
    n_split += int(split)
    st = eng.stats()
    Fr, dr = np.zeros((n, D), np.float64), np.zeros(n, np.float64)
    info = orc.backproject_view(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(), vm.numpy(), K.numpy(), W, H,
                                feats.numpy(), Fr, dr)
    ok = st["overflow"] == 0 and st["n_pairs"] == info["n_pairs"] and st["n_visible"] == info["n_vis"]
    eF = rel_row_err(F.cpu().numpy(), Fr) if info["n_pairs"] else 0.0
    ed = rel_row_err(d.cpu().numpy()[:, None], dr[:, None]) if info["n_pairs"] else 0.0
    ok = ok and eF <= 1e-4 and ed <= 1e-4 and (tight or st["n_isect"] == info["n_isect"])
    # the forward render of a random colour table over the same view (gwbp_render: k_render_rows for D_r < 128 or D_r % 4 != 0,
    # k_render_rows4 with 256 / 512 channels per wave otherwise) against the oracle's
    if ok and rng.random() < 0.5:
        Dr = int(rng.choice([5, 20, 64, 127, 128, 132, 200, 256, 260, 384, 512, 516, 708, 1024]))
        cols = torch.randn(n, Dr, generator=g)
        if fused or enc is not None or tok:  # those paths left no weight store behind
            eng.blend_weights(view)
            if eng.stats()["overflow"] & 2:  # (they never needed the pair capacity either: a store that does not fit here says
                continue                     # nothing about the render -- next case)
        out = eng.render(view, cols.to(dev)).cpu().numpy()
        rp = orc.project(means.numpy(), quats.numpy(), scales.numpy(), vm.numpy(), K.numpy(), W, H)
        rb = orc.bin_sort(rp, W, H)
        ref, _ = orc.render(rp, rb, opac.numpy(), cols.numpy(), W, H)
        er = float(np.abs(out - ref).max()) / max(1.0, float(np.abs(ref).max()))
        n_render += 1
        if eng.stats()["overflow"] or er > 2e-5:
            ok = False
            print(f"render D_r={Dr}: max error {er:.2e} overflow {eng.stats()['overflow']}", flush=True)
    # the pixel-parallel render (gwbp_render_pixels, 1..32 channels, no weight store): colours and the alpha map (bit for bit)
    if ok and rng.random() < 0.4:
        Dp = int(rng.choice([1, 3, 4, 5, 9, 16, 17, 32]))
        cols = torch.randn(n, Dp, generator=g)
        out, alpha = eng.render_pixels(view, cols.to(dev))
        rp = orc.project(means.numpy(), quats.numpy(), scales.numpy(), vm.numpy(), K.numpy(), W, H)
        rb = orc.bin_sort(rp, W, H)
        ref, ralpha = orc.render(rp, rb, opac.numpy(), cols.numpy(), W, H)
        er = float(np.abs(out.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        same_alpha = np.array_equal(alpha.cpu().numpy().view(np.uint32), ralpha.view(np.uint32))
        n_px += 1
        if er > 2e-5 or not same_alpha:
            ok = False
            print(f"render_pixels D={Dp}: max error {er:.2e} alpha identical {same_alpha}", flush=True)
    if not ok:
        bad += 1
        print(f"FAIL case {seed0 + case}: N={n} {W}x{H} D={D} s0={s0:.4f} {layout} up={up} wide={wide} tight={tight} fused={fused} tok={tok} split={split} enc={None if enc is None else tuple(enc.shape)} "
              f"pairs {st['n_pairs']}/{info['n_pairs']} eF={eF:.2e} ed={ed:.2e} overflow={st['overflow']}", flush=True)
print(f"{n_cases} cases ({n_fused} through the fused blend+scatter kernel, {n_enc} through the encoder-fused one ({n_split} of them in its producer / consumer form), {n_tok} through token space, {n_render} also rendered forward, {n_px} through the pixel-parallel render), {bad} failures, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
