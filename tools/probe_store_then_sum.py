"""Timing-only probe of a store-then-sum scatter AS A PIPELINE (round-3 review, item 4): what would the step cost if the flush
atomics were plain stores of partial rows and a later pass summed them into F?  Results are INVALID by design.

  scatter side   a PROFILE / ablation library whose flushes are plain stores of the same rows (GWBP_ABLATE=1 for k_scatter_full,
                 tools/lib/libgwbp_abl64.so for k_scatter_wide)
  sum pass       tools/lib/libsumpass.so (tools/ubench_sum_pass.hip -DSUMPASS_LIB): a synthetic index of the config's shape
                 (records / Gaussians with weight of view 0, profiles/r3_flush_cache_sim.txt), launched per view on a side stream
                 behind scatter(v) and beside scatter(v + 1), reading its own partial-row buffer and read-modify-writing the REAL F

usage (on the GPU box): [GWBP_ABLATE=1] python tools/probe_store_then_sum.py C4|C2 <views> sum|nosum [library]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import gsbp_amd  # noqa: E402
if len(sys.argv) > 4:
    gsbp_amd._lib.use_library(sys.argv[4], allow_profile=True)
from gsbp_amd import synthetic as syn  # noqa: E402

name, n_views, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
SHAPES = {"C4": (3721145, 1723388), "C2": (1810029, 585522)}
cfg = syn.CONFIGS[name]
dev = torch.device("cuda:0")
means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=n_views).to(dev), syn.intrinsics(cfg).to(dev)
N, W, H, D = cfg.n_gaussians, cfg.width, cfg.height, cfg.feat_dim
pool = [syn.make_feature_map(cfg, i, device=dev, dim=D) for i in range(2)]
side = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()
lib = None
if mode == "sum":
    lib = C.CDLL(os.path.abspath("tools/lib/libsumpass.so"))
    lib.sum_pass_setup.argtypes = [C.c_int, C.c_long, C.c_long, C.c_long]
    lib.sum_pass_launch.argtypes = [C.c_void_p, C.c_void_p]
    n_rec, n_dst = SHAPES[name]
    assert lib.sum_pass_setup(D, N, n_dst, n_rec) == 0
state = {"F": None, "calls": 0}


def feature_fn(v):
    # called on the scatter stream once per view, after scatter(v - 1) was enqueued there: the pass that sums view v - 1's rows
    # goes on the side stream behind it and runs beside scatter(v)
    if lib is not None and state["calls"] > 0 and state["F"] is not None:
        ev.record(torch.cuda.current_stream(dev))
        side.wait_event(ev)
        assert lib.sum_pass_launch(C.c_void_p(state["F"].data_ptr()), C.c_void_p(side.cuda_stream)) == 0
    state["calls"] += 1
    return pool[v % len(pool)]


# create_feature_field allocates its own accumulators: the pass read-modify-writes a stand-in F of the same size (same traffic)
F_standin = torch.zeros(N, D, device=dev)
state["F"] = F_standin
for rep in range(2):  # the first pass warms the workspaces up (capacity growth, scatter-kernel choice)
    state["calls"] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = gsbp_amd.create_feature_field(means, quats, scales, opac, vms, K, W, H, feature_fn, D, return_partials=True)
    side.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_views * 1e3
    del out
print(f"{name} {mode:6s} lib={os.path.basename(sys.argv[4]) if len(sys.argv) > 4 else 'product'} ablate={os.environ.get('GWBP_ABLATE', '-')}: "
      f"{dt:.3f} ms/view over {n_views} views (second pass)")
