#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

metric  : Gaussian-pixel-features/sec = sum over timed views of (#contributing (Gaussian, pixel) pairs) x D / time
workload: BASELINE.json configs[1] ("C2"): 1M synthetic Gaussians, 1600x1060 views, D = 512 feature maps
step    : one view of the hot path: project -> bin/sort -> blend weights -> scatter-accumulate into F[N,D], d[N]
          (backproject.py:115-151), inputs resident in HBM.  Views shard over ranks (r, r+R, ...); after the last
          step the ranks' partial F/d are summed with ONE all-reduce (RCCL over xGMI) inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C1|C4|C5] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the two-stream view pipeline needs its streams on different hardware queues (default 4; RCCL takes some): see
# ViewPipeline.  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--pool", type=int, default=4, help="feature maps cycled through (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=1)
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for 1 rank (test)")
    ap.add_argument("--exact-binning", action="store_true",
                    help="gsplat's 3-sigma tile binning instead of GWBP_FLAG_TIGHT_BINNING (same F and d either way)")
    ap.add_argument("--serial", action="store_true", help="one stream, no overlap of front(v+1) with scatter(v)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import gsbp_amd
    from gsbp_amd import synthetic as syn

    cfg = syn.CONFIGS[args.config]
    N, W, H = cfg.n_gaussians, cfg.width, cfg.height
    D_in = cfg.feat_dim
    means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
    K = syn.intrinsics(cfg)
    n_views_needed = world * (args.steps + args.warmup)
    vms = syn.make_cameras(cfg, n_views=max(cfg.n_views, n_views_needed))
    encoder = syn.make_encoder(cfg).to(dev) if cfg.encoder_dim else None
    D = cfg.encoder_dim or D_in

    # feature-map pool, generated on device (seeded), L2-normalised over channels like backproject.py:109
    pool = [syn.make_feature_map(cfg, 1000 * rank + i, device=dev) for i in range(args.pool)]
    tight = not args.exact_binning
    eng = gsbp_amd.Engine(N, W, H, device=dev, tight_binning=tight)
    F = torch.zeros(N, D, device=dev)
    d = torch.zeros(N, device=dev)
    my_views = [rank + world * i for i in range(args.steps + args.warmup)]
    views = [eng.view(vms[v], K, W, H) for v in my_views]

    # capacity check on one untimed view (the timed loop never reads sizes back)
    while True:
        eng.backproject_view(views[0], means, quats, scales, opac, pool[0] if encoder is None else pool[0] @ encoder,
                             F, d)
        st = eng.stats()
        if not st["overflow"]:
            break
        eng.grow(st)
    if args.serial:
        eng.set_narrow_scatter(not (D % 256 == 0 and "GWBP_NO_WIDE" not in os.environ))
        pipe, accum = None, torch.zeros(32, dtype=torch.uint8, device=dev)
    else:
        eng2 = gsbp_amd.Engine(N, W, H, device=dev, isect_cap=eng.isect_cap, pair_cap=eng.pair_cap, tight_binning=tight)
        pipe = gsbp_amd.ViewPipeline(N, W, H, dev, engines=[eng, eng2], scatter_dim=D)
        accum = pipe.accum

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    n_total = args.steps + args.warmup

    def front(i):
        """project -> bin/sort -> blend of view i on the side stream (overlaps scatter of view i-1)."""
        k = i - args.warmup
        if not args.serial:
            if 0 <= k < args.steps:
                with torch.cuda.stream(pipe.side):
                    ev[k][0].record(pipe.side)
            pipe.front(views[i], means, quats, scales, opac, d)  # d: added behind the blend when the wide kernel is used
            if 0 <= k < args.steps:
                with torch.cuda.stream(pipe.side):
                    ev[k][1].record(pipe.side)

    def scatter(i):
        k = i - args.warmup
        feats = pool[i % args.pool]
        if encoder is not None:
            feats = feats @ encoder  # backproject_compressed.py:127 happens inside the timed step
        if args.serial:  # one stream, one workspace: the pre-pipelining schedule
            eng.project(views[i], means, quats, scales, opac)
            eng.bin_sort(views[i])
            eng.blend_weights(views[i])
            if 0 <= k < args.steps:
                ev[k][2].record()
            eng.scatter(views[i], feats, F, d)
            eng.accumulate_stats(accum)
            if 0 <= k < args.steps:
                ev[k][3].record()
            return
        timed = 0 <= k < args.steps
        pipe.scatter(feats, F, d, t0=ev[k][2] if timed else None, t1=ev[k][3] if timed else None)

    front(0)
    for i in range(args.warmup):
        front(i + 1)
        scatter(i)
    torch.cuda.synchronize(dev)
    scatter_choice = "narrow"
    if not args.serial:  # the warm-up views' counters pick the scatter kernel (256- or 128-channel) for the timed ones
        st_w = gsbp_amd.Engine.decode_stats(accum)
        scatter_choice = pipe.choose_scatter_kernel(st_w["n_pairs"], st_w["n_headers"])
    elif D % 256 == 0 and "GWBP_NO_WIDE" not in os.environ:
        scatter_choice = "wide"  # serial schedule: the faster kernel alone (set before the warm-up), no priority
    F.zero_()
    d.zero_()
    accum.zero_()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if use_dist:  # first use of each collective (communicator channels, staging buffers) stays outside the timed region
        wf = torch.zeros(world * 8, D, device=dev)
        gsbp_amd.reduce_partials_sharded(wf, torch.zeros(world * 8, device=dev))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        if i + 1 < n_total:
            front(i + 1)
        scatter(i)
    if use_dist:
        # the path's one exchange step, inside the timed region: reduce-scatter of F (rank r keeps the rows it would
        # finalise), all-reduce of d
        F_rows, d_rows, row0 = gsbp_amd.reduce_partials_sharded(F, d)
    barrier()
    elapsed = time.perf_counter() - t0

    stats = gsbp_amd.Engine.decode_stats(accum)
    tt = torch.tensor([elapsed, float(stats["n_pairs"]), float(stats["overflow"])], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_pairs, overflow = float(tt[1]), float(tmax[2])
    else:
        total_pairs, overflow = float(tt[1]), float(tt[2])

    # front(k) of the first timed view was enqueued during warm-up; its events are not recorded
    fr = [e[0].elapsed_time(e[1]) for e in ev[1:]] if (not args.serial and args.steps > 1) else [0.0]
    t_front = sum(fr) / len(fr)
    t_scatter = sum(e[2].elapsed_time(e[3]) for e in ev) / args.steps

    if rank == 0:
        n_vis = stats["n_visible"] / args.steps
        n_isect = stats["n_isect"] / args.steps
        n_hdr = stats["n_headers"] / args.steps
        pairs_view = stats["n_pairs"] / args.steps
        # algorithmic bytes of ONE scatter launch: feature map read once + RMW of the F rows
        # and d entries of the Gaussians visible in the view (4HWD + 8 N_vis (D+1))
        b_scatter = 4.0 * H * W * D + 8.0 * n_vis * (D + 1)
        b_view = b_scatter + 44.0 * N + 24.0 * n_isect
        achieved = b_scatter / (t_scatter * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.config, {}).get("scatter_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        scatter_kernel = ("k_scatter_wide" if scatter_choice == "wide" else
                          "k_scatter_full" if (D % 128 == 0 or D <= 64) else "k_scatter")
        out = {
            "metric": "Gaussian-pixel-features/sec", "value": total_pairs * D / elapsed,
            "unit": "Gaussian-pixel-features/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {N} Gaussians, {W}x{H} views, D={D_in}"
                                   + (f"->{D} (encoder)" if encoder is not None else "")
                                   + f", {args.steps} views/GPU, view-sharded over {world} GPU(s), one reduce-scatter of F + all-reduce of d",
                       "views_per_sec": world * args.steps / elapsed, "pairs_per_view": pairs_view,
                       "n_visible_per_view": n_vis, "n_isect_per_view": n_isect, "n_headers_per_view": n_hdr,
                       "binning": "alpha-ellipse bounding box (GWBP_FLAG_TIGHT_BINNING)" if tight else "gsplat 3-sigma square",
                       "overflow": overflow,
                       "schedule": "serial" if args.serial else "front(v+1) overlapped with scatter(v) on two streams",
                       "stage_ms": {"front(project+sort+blend, side stream, overlapped)": t_front,
                                    "scatter": t_scatter}},
            "roofline": {"bound": "hbm", "kernel": scatter_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_scatter, "launch_ms": t_scatter,
                         "pipeline_achieved_GBs": b_view / (elapsed / args.steps) / 1e9,
                         # second ceiling of the same kernel: fp32 atomics execute memory-side at ~1.3 TB/s of added
                         # bytes chip-wide (MI355X_MICROARCH.md, Global float atomics); one flush per (Gaussian, tile)
                         "atomic_added_GBs": n_hdr * (D + 1) * 4.0 / (t_scatter * 1e-3) / 1e9,
                         "atomic_peak_GBs": 1300.0,
                         "atomic_frac": n_hdr * (D + 1) * 4.0 / (t_scatter * 1e-3) / 1e9 / 1300.0},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, syn, means, quats, scales, opac, vms, K, pool, D, encoder,
                                               args.cpu_views)
        line = json.dumps(out)
    else:
        line = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(line, flush=True)  # the ONE JSON line, after any RCCL teardown chatter
    if use_dist:
        # RCCL prints its version banner from a library destructor when NCCL_DEBUG=VERSION is set (it is on the GPU
        # boxes), i.e. AFTER this point and on stdout.  Leave without running destructors so that the JSON line stays
        # the last line of stdout; everything is flushed and the process group is already destroyed.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def cpu_baseline(cfg, syn, means, quats, scales, opac, vms, K, pool, D, encoder, n_views):
    """The oracle (a CPU port of the same algorithm; the reference itself has no CPU rasteriser) timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    import numpy as np
    cores = os.cpu_count() or 1
    h = [t.cpu().numpy() for t in (means, quats, scales, opac)]
    Fc = np.zeros((cfg.n_gaussians, D), np.float32)
    dc = np.zeros(cfg.n_gaussians, np.float32)
    feats = [(p if encoder is None else p @ encoder).cpu().numpy() for p in pool[:n_views]]
    orc.lib()
    pairs, t = 0, 0.0
    for v in range(n_views):
        t0 = time.perf_counter()
        info = orc.backproject_view(h[0], h[1], h[2], h[3], vms[v].numpy(), K.numpy(), cfg.width, cfg.height,
                                    feats[v % len(feats)], Fc, dc, nthreads=cores)
        t += time.perf_counter() - t0
        pairs += info["n_pairs"]
    return {"value": pairs * D / t, "unit": "Gaussian-pixel-features/s", "cores": cores, "kind": "port",
            "sample": f"{n_views} view(s) of {cfg.name} at full size (N={cfg.n_gaussians}, {cfg.width}x{cfg.height}, "
                      f"D={D}), oracle/gwbp_oracle.c with OpenMP on all host cores, fp32 accumulators",
            "seconds": t, "views_per_sec": n_views / t}


if __name__ == "__main__":
    main()
