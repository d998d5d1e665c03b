"""Scatter time at C2 geometry for the three feature-map forms: [H,W,D] contiguous, channel-major [D,H,W] view,
low-resolution [h,w,D] with nearest upsampling inside the kernel.  GPU only."""
import sys
import torch
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn

dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C2"]
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
vms, K = syn.make_cameras(cfg, n_views=1), syn.intrinsics(cfg)
view = eng.view(vms[0], K, cfg.width, cfg.height)
eng.project(view, *g); eng.bin_sort(view); eng.blend_weights(view)
low = torch.randn(64, 64, D, device=dev)
planar = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(cfg.height, cfg.width), mode="nearest")[0]
low240 = torch.randn(240, 240, D, device=dev)
forms = {"bilinear 240x240": (low240, "bilinear"), "lowres+maps": (low, "nearest"), "channel-major": (planar.permute(1, 2, 0), None),
         "contiguous": (planar.permute(1, 2, 0).contiguous(), None)}
F = torch.zeros(cfg.n_gaussians, D, device=dev)
d = torch.zeros(cfg.n_gaussians, device=dev)
res = {}
for name, (m, up) in forms.items():
    F.zero_(); d.zero_()
    for _ in range(2):
        eng.scatter(view, m, F, d, upsample=up)
    F.zero_(); d.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.scatter(view, m, F, d, upsample=up)
    e1.record(); torch.cuda.synchronize()
    res[name] = F.clone()
    print(f"D={D} {name:14s} {e0.elapsed_time(e1) / 5:.3f} ms/scatter", flush=True)
ref = res["contiguous"]
for name in res:
    if name.startswith("bilinear"):
        continue
    print(name, "max rel diff vs contiguous", float((res[name] - ref).norm(dim=1).max() / ref.norm(dim=1).max()))
