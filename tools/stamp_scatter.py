"""Where does k_scatter_wide spend its wave-time?  PROFILE library only (make -C <pkg>/csrc PROFILE=1):
python tools/stamp_scatter.py [views]"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import torch, gsbp_amd
from gsbp_amd import _lib
_lib.use_library(os.path.abspath("tools/lib/libgwbp_profile.so"), allow_profile=True)
from gsbp_amd import synthetic as syn
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C2"]
V = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=V), syn.intrinsics(cfg)
eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
eng.set_narrow_scatter(False)
feats = syn.make_feature_map(cfg, 0, device=dev)
F = torch.zeros(cfg.n_gaussians, cfg.feat_dim, device=dev); d = torch.zeros(cfg.n_gaussians, device=dev)
L = gsbp_amd.lib(); buf = (C.c_ulonglong * 8)()
for v in range(V):
    view = eng.view(vms[v], K, cfg.width, cfg.height)
    eng.project(view, *g); eng.bin_sort(view); eng.blend_weights(view)
    torch.cuda.synchronize()
    if v == 1: L.gwbp_profile_read_wide(buf)  # reset after warm-up
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.scatter(view, feats, F, None); e1.record(); torch.cuda.synchronize()
    if v >= 1: print("view", v, "scatter ms %.3f" % e0.elapsed_time(e1))
L.gwbp_profile_read_wide(buf)
tot = sum(buf[i] for i in range(4))
names = ["table+commit+barrier", "visit loop", "next-item wait", "round barrier wait"]
for i in range(4):
    print("%-18s %5.1f %%" % (names[i], 100.0 * buf[i] / tot))
print("phases(waves) %d visits %d -> cycles/visit in loop %.0f; per view wave-cycles %.3g" % (buf[4], buf[5], buf[1] / max(1, buf[5]), tot / (V - 1)))
