"""C2-size timing of (a) the wide-D forward render and (b) the reference's own per-view loop run through the drop-in
rasterization() shim (two rasterise+backward passes per view, backproject.py:115-151) vs the fused call.  GPU only."""
import sys
import time
import torch
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn
from gsbp_amd import rasterization

dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C2"]
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=3), syn.intrinsics(cfg)
vms, K = vms.to(dev), K.to(dev)
W, H, N = cfg.width, cfg.height, cfg.n_gaussians
feats = syn.make_feature_map(cfg, 0, device=dev, dim=D)


def sync_time(fn, n=5):
    for _ in range(3):  # the caching allocator settles only after a few iterations of 3.5 GB temporaries
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


# (a) forward render of a D-wide colour table
colors = torch.randn(N, D, device=dev)
eng = gsbp_amd.Engine(N, W, H, device=dev)
view = eng.view(vms[0].cpu(), K.cpu(), W, H)
eng.project(view, means, quats, scales, opac); eng.bin_sort(view); eng.blend_weights(view)
print(f"k_render forward D={D}: {sync_time(lambda: eng.render(view, colors)):.2f} ms", flush=True)
del colors

# (b) the reference loop through the shim
colors_feats = torch.zeros(N, D, device=dev, requires_grad=True)
colors_0 = torch.zeros(N, 3, device=dev, requires_grad=True)
gf = torch.zeros(N, D, device=dev)
gd = torch.zeros(N, device=dev)


calls = [0]


def ref_view():
    v = calls[0] % vms.shape[0]  # a NEW view per call, as in the job: the shim's front stage (project, sort, blend) runs once
    calls[0] += 1                # per view; the second rasterization() of the view finds its result in the workspace
    out, _, _ = rasterization(means, quats, scales, opac, colors_feats, vms[v][None], K[None], width=W, height=H)
    (out[0] * feats).sum().backward()
    colors_feats_copy = colors_feats.grad.clone()  # (the reference's own statements, backproject.py:127-151, in its order)
    colors_feats.grad.zero_()
    out, _, _ = rasterization(means, quats, scales, opac, colors_0, vms[v][None], K[None], width=W, height=H)
    out[0].sum().backward()
    gf.add_(colors_feats_copy)
    gd.add_(colors_0.grad[:, 0])
    colors_0.grad.zero_()


print(f"reference loop through the shim: {sync_time(ref_view):.2f} ms/view", flush=True)

F = torch.zeros(N, D, device=dev); d = torch.zeros(N, device=dev)
eng2 = gsbp_amd.Engine(N, W, H, device=dev, tight_binning=True)
eng2.set_narrow_scatter(D % 256 != 0)  # the kernel ViewPipeline picks at this D (an Engine starts narrow)
print(f"fused backproject_view: {sync_time(lambda: eng2.backproject_view(view, means, quats, scales, opac, feats, F, d)):.2f} ms/view")
