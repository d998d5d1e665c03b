"""The 256-channel scatter kernel against the 128-channel one on the same views (T1, C1 at D = 256, C2 at D = 512), optionally
with another build of the library: python tools/wide_vs_narrow.py [tools/lib/libgwbp_<name>.so].  Exit code 1 on a mismatch.
Round 5 found a latent miscompilation hazard of k_scatter_wide this way (a PROFILE + in-kernel-stamps build copied the landing
registers of two in-flight loads in front of their wait): tests/test_gpu_parity.py runs it on the PROFILE library when that
has been built (make -C <pkg>/csrc PROFILE=1)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gsbp_amd
from gsbp_amd import _lib, synthetic as syn
if len(sys.argv) > 1:
    _lib.use_library(sys.argv[1], allow_profile=True)
from util import scene_np, to_dev
dev = torch.device("cuda:0")
for name, D in (("T1", 256), ("C1", 256), ("C2", 512)):
    cfg, sc = scene_np(name)
    d = to_dev(sc, dev)
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    f = syn.make_feature_map(cfg, 0, dim=D).to(dev)
    res = []
    for wide in (True, False):
        eng = gsbp_amd.Engine(N, W, H, device=dev, tight_binning=True)
        eng.set_narrow_scatter(not wide)
        view = eng.view(d["vms"][0], d["K"], W, H)
        eng.project(view, d["means"], d["quats"], d["scales"], d["opac"]); eng.bin_sort(view); eng.blend_weights(view)
        F = torch.zeros(N, D, device=dev)
        eng.scatter(view, f, F, None)
        torch.cuda.synchronize()
        res.append(F)
    bad = (~torch.isfinite(res[0]).all(dim=1))
    diff = (res[0] - res[1]).norm(dim=1).max() / res[1].norm(dim=1).max()
    print(name, D, "non-finite rows in wide:", int(bad.sum()), "max rel diff wide vs narrow:", float(diff), flush=True)
    worst = max(globals().get("worst", 0.0), float(diff))
    if int(bad.sum()):
        worst = float("inf")
        rows = torch.nonzero(bad)[:5, 0].tolist()
        print("  rows", rows, "channels non-finite in first row:", torch.nonzero(~torch.isfinite(res[0][rows[0]]))[:8, 0].tolist())

sys.exit(0 if worst <= 1e-5 else 1)
