"""C2-size timing of the RGB forward render (`gwbp_render_pixels`, D <= 32: backproject.py:89-100 renders the view it feeds to the 2-D network
this way) and of the SH colour evaluation in front of it; project + sort once, then each kernel REPS times.  GPU only.
usage: python3 tools/time_render_px.py [D] [reps] [config] [library]"""
import sys
import torch
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn

dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = syn.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "C2"]
if len(sys.argv) > 4:
    gsbp_amd._lib.use_library(sys.argv[4], allow_profile=True)
means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=2), syn.intrinsics(cfg)
W, H, N = cfg.width, cfg.height, cfg.n_gaussians
colors = torch.rand(N, D, generator=torch.Generator().manual_seed(3)).to(dev)
eng = gsbp_amd.Engine(N, W, H, device=dev, tight_binning=True)
view = eng.view(vms[0], K, W, H)
eng.project(view, means, quats, scales, opac)
eng.bin_sort(view)
st = eng.stats()
assert not st["overflow"]


def timed(fn):
    for _ in range(2):
        r = fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        r = fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps, r


ms, (out, alpha) = timed(lambda: eng.render_pixels(view, colors))
print(f"render_pixels D={D}: {ms:.3f} ms  intersections {st['n_isect']}  checksum {float(out.double().sum()):.6e} alpha {float(alpha.double().sum()):.6e}")
ms_b, _ = timed(lambda: eng.blend_weights(view))
print(f"blend_weights (for comparison: the same per-pixel arithmetic + the weight store): {ms_b:.3f} ms")
sh = torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(4)).to(dev)
ms_s, _ = timed(lambda: eng.sh_colors(3, means, sh, [0.0, 0.0, 3.0]))
print(f"sh_colors degree 3: {ms_s:.3f} ms")
