"""Scatter time at C2 (view 0) for the 256- and the 128-channel kernel, with and without the denominator accumulation
(d = None).  GPU only."""
import sys
import torch
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn

dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C2"]
D = 512
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
vms, K = syn.make_cameras(cfg, n_views=1), syn.intrinsics(cfg)
view = eng.view(vms[0], K, cfg.width, cfg.height)
wide = "--narrow" not in sys.argv
eng.set_narrow_scatter(not wide)  # before the blend: the wide kernel needs its half-tile lists
eng.project(view, *g); eng.bin_sort(view); eng.blend_weights(view)
print("kernel:", "k_scatter_wide (+ k_accum_d)" if wide else "k_scatter_full")
feats = syn.make_feature_map(cfg, 0, device=dev)
F = torch.zeros(cfg.n_gaussians, D, device=dev)
d = torch.zeros(cfg.n_gaussians, device=dev)
for name, dd in (("with d", d), ("without d", None), ("with d", d), ("without d", None)):
    for _ in range(2):
        eng.scatter(view, feats, F, dd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.scatter(view, feats, F, dd)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 10:.3f} ms/scatter", flush=True)
print(eng.stats())
