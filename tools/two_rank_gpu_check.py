"""Two ranks sharing one GPU (gloo backend, so no RCCL duplicate-device error): the view-sharded driver with the real HIP
engines in every rank against the single-process sum.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29566 tools/two_rank_gpu_check.py"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, ".")
import gsbp_amd
from gsbp_amd import synthetic as syn
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C1"]
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=8).to(dev), syn.intrinsics(cfg).to(dev)
D = 256
fn = lambda v: syn.make_feature_map(cfg, v, device=dev, dim=D)
out, F, d, st = gsbp_amd.create_feature_field(*g, vms, K, cfg.width, cfg.height, fn, D, return_partials=True)
# single-process reference on rank 0: all views, no process group semantics -> use explicit views list and manual sum
if rank == 0:
    import types
    F1 = torch.zeros_like(F); d1 = torch.zeros_like(d)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    for v in range(8):
        eng.backproject_view(eng.view(vms[v].cpu(), K.cpu(), cfg.width, cfg.height), *g, fn(v), F1, d1)
    torch.cuda.synchronize()
    print("rank0 views", syn.view_shard(8, 0, world), "max rel F diff", float(((F - F1).norm(dim=1) / F1.norm(dim=1).clamp_min(1e-6 * float(F1.norm(dim=1).max()))).max()),
          "d diff", float((d - d1).abs().max() / d1.max()))
dist.barrier()
dist.destroy_process_group()
