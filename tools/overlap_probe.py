#!/usr/bin/env python3
"""How do k_encode_map (HBM streaming) and the fused blend+scatter kernel (vector-issue-bound) slow each other down?
Each kernel is launched back to back on its own stream, alone and together; durations from events on the launch streams.
usage: python tools/overlap_probe.py [encoder workgroups per CU ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsbp_amd  # noqa: E402
from gsbp_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C5"]
g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
vms, K = syn.make_cameras(cfg, n_views=1), syn.intrinsics(cfg)
enc = syn.make_encoder(cfg).to(dev)
fmap = syn.make_feature_map(cfg, 0, device=dev)
eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
view = eng.view(vms[0], K, cfg.width, cfg.height)
F = torch.zeros(cfg.n_gaussians, cfg.encoder_dim, device=dev)
d = torch.zeros(cfg.n_gaussians, device=dev)
small = eng.encode_map(fmap, enc)
while True:
    eng.project(view, *g)
    eng.bin_sort(view)
    eng.blend_scatter(view, small, F, d)
    st = eng.stats()
    if not st["overflow"]:
        break
    eng.grow(st)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
sa, sb = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
REP = 20


def timed(stream, fn, rep=REP):
    evs = []
    with torch.cuda.stream(stream):
        for _ in range(rep):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            evs.append((e0, e1))
    return evs


def med(evs):
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


for per_cu in [float(x) for x in sys.argv[1:]] or [1.0, 2.0, 4.0]:
    wg = max(1, int(per_cu * n_cu))
    enc_fn = lambda: eng.encode_map(fmap, enc, workgroups=wg)
    bl_fn = lambda: eng.blend_scatter(view, small, F, d)
    torch.cuda.synchronize()
    a = timed(sa, enc_fn)
    torch.cuda.synchronize()
    b = timed(sb, bl_fn)
    torch.cuda.synchronize()
    # together: the encoder stream is kept busy for the whole duration of the blend launches and vice versa
    ea = timed(sa, enc_fn, 3 * REP)
    eb = timed(sb, bl_fn, REP)
    torch.cuda.synchronize()
    print(f"encoder {per_cu:4.2f} WG/CU: alone {med(a):6.3f} ms, beside the blend {med(ea[:REP]):6.3f} | blend+scatter alone "
          f"{med(b):6.3f} ms, beside the encoder {med(eb):6.3f}")
