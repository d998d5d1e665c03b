"""C2-size timing of the wide forward render alone (`gwbp_render`: out[p,:] = sum_g w_g(p) colors[g,:], what segment.py:209-220 does with
the 512-d field for every frame): the front stage runs once, then the render kernel REPS times.  GPU only.
usage: python3 tools/time_render.py [D] [reps] [config] [library]"""
import sys
import torch
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn

dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = syn.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "C2"]
if len(sys.argv) > 4:
    gsbp_amd._lib.use_library(sys.argv[4], allow_profile=True)
means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=2), syn.intrinsics(cfg)
W, H, N = cfg.width, cfg.height, cfg.n_gaussians
colors = torch.randn(N, D, generator=torch.Generator().manual_seed(3)).to(dev)
eng = gsbp_amd.Engine(N, W, H, device=dev, tight_binning=True)
view = eng.view(vms[0], K, W, H)
eng.project(view, means, quats, scales, opac)
eng.bin_sort(view)
eng.blend_weights(view)
st = eng.stats()
assert not st["overflow"]
for _ in range(2):
    out = eng.render(view, colors)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    out = eng.render(view, colors)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / reps
alg = 4.0 * H * W * D + 4.0 * st["n_headers"] * D  # write the image once + read every record's colour row once
print(f"render {cfg.name if hasattr(cfg, 'name') else ''} D={D}: {ms:.3f} ms  records {st['n_headers']}  pairs {st['n_pairs']}  "
      f"algorithmic {alg / 1e9:.2f} GB = {alg / ms / 1e6 / 8000:.3f} of 8 TB/s  checksum {float(out.double().sum()):.6e}")
