"""N > 1 path on CPU: world_size-2 gloo process group, views sharded r, r+R, ..., one exchange step (reduce-scatter of
F on a row count padded to the world size + all-reduce of d), row-local finalise of each rank's block, all-gather of the
finalised blocks -- must equal the single-process result.  N = 301 does not divide by 2."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gsbp_amd
    from gsbp_amd import synthetic as syn
    from oracle import oracle as orc
    cfg = syn.Config("D0", 301, 5, 80, 48, 6, 0.07, False)
    means, quats, scales, opac = syn.activate(syn.make_scene(cfg))
    K, vms = syn.intrinsics(cfg), syn.make_cameras(cfg)
    h = [t.numpy() for t in (means, quats, scales, opac)]
    F = torch.zeros(cfg.n_gaussians, cfg.feat_dim)
    d = torch.zeros(cfg.n_gaussians)
    seen = []
    per = -(-cfg.n_gaussians // world)

    def view_fn(v, feats):
        seen.append(v)
        Fv = np.zeros((cfg.n_gaussians, cfg.feat_dim), np.float32)
        dv = np.zeros(cfg.n_gaussians, np.float32)
        orc.backproject_view(*h, vms[v].numpy(), K.numpy(), cfg.width, cfg.height, feats.numpy(), Fv, dv, nthreads=1)
        F.add_(torch.from_numpy(Fv))
        d.add_(torch.from_numpy(dv))

    # the product driver with the per-view kernel call replaced by the oracle: sharding, padded reduce-scatter,
    # row-local finalise and all-gather are the code that ships.  (view_fn accumulates into this test's own F/d; the
    # driver's internal accumulators stay zero, so only its plumbing is exercised here and checked further down.)
    drv = {}

    def view_fn_drv(v, feats):
        view_fn(v, feats)
        drv["F"][:] = F  # mirror into the driver's (padded-storage) accumulators
        drv["d"][:] = d

    orig_alloc = gsbp_amd.backproject.alloc_accumulators

    def spy_alloc(n, dim, device, w=None):
        Fa, da, store = orig_alloc(n, dim, device, w)
        assert store.shape[0] == per * world and Fa.data_ptr() == store.data_ptr()
        drv["F"], drv["d"] = Fa, da
        return Fa, da, store

    gsbp_amd.backproject.alloc_accumulators = spy_alloc
    out_full, F_rows_drv, d_sum_drv, st = gsbp_amd.create_feature_field(
        means, quats, scales, opac, vms, K, cfg.width, cfg.height, lambda v: syn.make_feature_map(cfg, v),
        cfg.feat_dim, view_fn=view_fn_drv, return_partials=True)
    gsbp_amd.backproject.alloc_accumulators = orig_alloc
    assert seen == syn.view_shard(cfg.n_views, rank, world)
    assert out_full.shape == (cfg.n_gaussians, cfg.feat_dim) and st["row0"] == rank * per
    # the sharded form of the exchange (what bench.py times): rank r gets its block of rows of the summed F and d
    F_rows, d_rows, row0 = gsbp_amd.reduce_partials_sharded(F.clone(), d.clone())
    gsbp_amd.reduce_partials(F, d)  # the single exchange step
    assert torch.equal(F_rows, F[row0:row0 + F_rows.shape[0]]) and torch.equal(d_rows, d[row0:row0 + d_rows.shape[0]])
    assert row0 == rank * per and F_rows.shape[0] == min(per, cfg.n_gaussians - row0)
    out = gsbp_amd.finalize_reference(F, d)
    # the driver's own exchange (in-place reduce-scatter on the padded storage) gives the same rows, sums and field
    assert torch.equal(F_rows_drv, F_rows) and torch.equal(d_sum_drv, d)
    assert torch.equal(out_full, out)
    if rank == 0:
        q.put((out_full.numpy(), F.numpy(), d.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_sharding_equals_single_process(orc):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, F, d = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import gsbp_amd  # noqa: F401
    from gsbp_amd import synthetic as syn
    cfg = syn.Config("D0", 301, 5, 80, 48, 6, 0.07, False)
    means, quats, scales, opac = [t.numpy() for t in syn.activate(syn.make_scene(cfg))]
    ref, Fr, dr, _ = orc.backproject_oracle(means, quats, scales, opac, syn.make_cameras(cfg).numpy(),
                                            syn.intrinsics(cfg).numpy(), cfg.width, cfg.height,
                                            lambda v: syn.make_feature_map(cfg, v).numpy(), cfg.feat_dim)
    assert np.abs(F - Fr).max() <= 1e-4 * max(1.0, np.abs(Fr).max())
    assert np.abs(out - ref).max() < 1e-5
