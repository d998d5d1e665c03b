"""Stage times of the CPU oracle on this host (thread-count sweep): python tools/time_oracle.py [C2] [D]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as orc
import gsbp_amd
from gsbp_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = syn.CONFIGS[name]
D = int(sys.argv[2]) if len(sys.argv) > 2 else (cfg.encoder_dim or cfg.feat_dim)
h = [t.numpy() for t in syn.activate(syn.make_scene(cfg))]
K, vm = syn.intrinsics(cfg).numpy(), syn.make_cameras(cfg, n_views=1).numpy()[0]
feats = syn.make_feature_map(cfg, 0, dim=D).numpy()
orc.lib()
for nt in (os.cpu_count(), 128, 64, 32, 16):
    if nt > (os.cpu_count() or 1):
        continue
    os.environ["OMP_NUM_THREADS"] = str(nt)
    F = np.zeros((cfg.n_gaussians, D), np.float32); d = np.zeros(cfg.n_gaussians, np.float32)
    t0 = time.perf_counter(); proj = orc.project(*h[:3], vm, K, cfg.width, cfg.height)
    t1 = time.perf_counter(); bins = orc.bin_sort(proj, cfg.width, cfg.height)
    t2 = time.perf_counter(); n, _ = orc.blend_scatter(proj, bins, h[3], feats, F, d, cfg.width, cfg.height, nthreads=nt)
    t3 = time.perf_counter()
    print(f"{name} D={D} threads={nt:3d}: project {t1-t0:6.3f}  bin/sort {t2-t1:6.3f}  blend+scatter {t3-t2:6.3f}  pairs {n}", flush=True)
