"""Worker of tests/test_gpu_two_ranks.py: one of two ranks that share ONE GPU (gloo process group -- RCCL refuses two
ranks on one device), each with its own HIP engines, running the view-sharded product driver: sharding r, r+R, ...,
the padded reduce-scatter of F + all-reduce of d, row-local finalise, all-gather.  Rank 0 compares with the
single-process result.  Launched by torch.distributed.run as a FRESH process (nothing here re-execs)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsbp_amd  # noqa: E402
from gsbp_amd import synthetic as syn  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
cfg = syn.Config("R2", 9999, 8, 400, 300, 256, 0.03, False)  # N does not divide by 2; D % 256 == 0: wide kernel
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg).to(dev), syn.intrinsics(cfg).to(dev)


def fn(v):
    return syn.make_feature_map(cfg, v, device=dev)


out, F_rows, d, st = gsbp_amd.create_feature_field(*g, vms, K, cfg.width, cfg.height, fn, cfg.feat_dim,
                                                   return_partials=True)
per = -(-cfg.n_gaussians // world)
assert st["overflow"] == 0 and st["row0"] == rank * per and out.shape == (cfg.n_gaussians, cfg.feat_dim)
assert F_rows.shape[0] == min(per, cfg.n_gaussians - rank * per)
if rank == 0:  # single-process sums over ALL views with a plain engine
    F1 = torch.zeros(cfg.n_gaussians, cfg.feat_dim, device=dev)
    d1 = torch.zeros(cfg.n_gaussians, device=dev)
    eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev)
    for v in range(cfg.n_views):
        eng.backproject_view(eng.view(vms[v].cpu(), K.cpu(), cfg.width, cfg.height), *g, fn(v), F1, d1)
    ref = eng.finalize(F1, d1)
    torch.cuda.synchronize()
    scale = float(F1.norm(dim=1).max())
    e_rows = float((F_rows - F1[:per]).norm(dim=1).max()) / scale
    e_d = float((d - d1).abs().max() / d1.max())
    e_out = float((out - ref).abs().max())
    ok = e_rows <= 2e-5 and e_d <= 2e-5 and e_out <= 1e-4
    print(f"TWO_RANK_{'OK' if ok else 'FAIL'} F_rows {e_rows:.2e} d {e_d:.2e} out {e_out:.2e}", flush=True)
dist.barrier()
dist.destroy_process_group()
