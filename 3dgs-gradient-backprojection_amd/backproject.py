"""Feature-field builder: counterpart of create_feature_field_lseg / _dino (backproject.py:25-172, :175-298)
and of the compressed variant (backproject_compressed.py:39-186) on top of the fused HIP path.

Per view the reference does: rasterise(zeros [N,D]) -> (render*feats).sum().backward() -> grad.clone();
rasterise(zeros [N,3]) -> render.sum().backward() -> grad[:,0]; F += ..; d += .. (backproject.py:115-151).
Here one call (`Engine.backproject_view`) projects, sorts, blends ONCE and scatter-accumulates straight into
F[N,D] and d[N].  Views shard across ranks (`synthetic.view_shard`); the partial F/d are summed in ONE exchange
step (a reduce-scatter of F + an all-reduce of the N-float d; RCCL over xGMI when the process group backend is
"nccl"), the 1e-12 of backproject.py:63 is added once after the reduction, normalisation is row-local
(backproject.py:166-169) so every rank finalises its own row block, and one all-gather of the finalised blocks
follows if the whole field is wanted on every rank.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Sequence

import torch

from .engine import Engine
from .synthetic import view_shard


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def reduce_partials(F: torch.Tensor, d: torch.Tensor) -> None:
    """The path's single exchange step: sum the per-rank partial accumulators."""
    dist, _, world = _dist()
    if world > 1:
        dist.all_reduce(F, op=dist.ReduceOp.SUM)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)


def encode_features(eng: Engine, feats: torch.Tensor, encoder: torch.Tensor, **kw) -> torch.Tensor:
    """backproject_compressed.py:127 for the DRIVER: the hand-written kernel (Engine.encode_map), whose domain -- [H,W,K]
    float32 maps with channel-contiguous 16-B aligned pixels, K % 16 == 0, K <= 2048, at most 16 outputs -- includes the
    reference's 512 -> 16 encoder.  Anything else RAISES (Engine.encode_map's own error): the drivers never switch to a
    library GEMM behind the caller's back.  Whoever wants one encodes in the feature function (`feats @ encoder`) and passes
    encoder=None."""
    return eng.encode_map(feats, encoder, **kw)


def rows_per_rank(n: int, world: int) -> int:
    """Rows of F every rank owns after the reduce-scatter: ceil(n / world); the last rank's block may be short."""
    return -(-n // world)


def alloc_accumulators(n: int, dim: int, device, world: Optional[int] = None):
    """F[n, dim], d[n] zero accumulators whose STORAGE is padded to a multiple of the world size, so that the
    reduce-scatter of reduce_partials_sharded runs in place on it (no 2 GB pad copy).  Returns (F, d, F_storage)."""
    if world is None:
        world = _dist()[2]
    n_pad = rows_per_rank(n, world) * world
    F_store = torch.zeros(n_pad, dim, device=device, dtype=torch.float32)
    return F_store[:n], torch.zeros(n, device=device, dtype=torch.float32), F_store


def reduce_partials_sharded(F: torch.Tensor, d: torch.Tensor, F_storage: Optional[torch.Tensor] = None):
    """The path's single exchange step as a reduce-scatter: rank r receives rows [row0, row0 + n_r) of the summed F (and
    the whole summed d, 4 B per Gaussian), which is all the row-local finalise needs (backproject.py:166-169) -- half
    the xGMI traffic of an all-reduce; an all-gather of the finalised rows follows only if one rank must hold the whole
    field.  Returns (F_rows, d_rows, row0).

    N need not divide by the world size: the reduce-scatter runs on ceil(N / world) * world rows -- on `F_storage` when
    the caller allocated F with alloc_accumulators (no copy), else on a zero-padded copy."""
    dist, rank, world = _dist()
    n = F.shape[0]
    if dist is None:  # no process group: nothing to exchange
        return F, d, 0
    # (a one-rank process group still goes through the collectives: `bench.py --force-dist` exercises the RCCL calls)
    per = rows_per_rank(n, world)
    row0, row1 = min(rank * per, n), min((rank + 1) * per, n)
    dist.all_reduce(d, op=dist.ReduceOp.SUM)
    src = F
    if per * world != n:
        if F_storage is not None and F_storage.shape[0] == per * world and F_storage.data_ptr() == F.data_ptr():
            src = F_storage
        else:
            src = torch.cat([F, F.new_zeros(per * world - n, F.shape[1])])
    out = torch.empty(per, F.shape[1], device=F.device, dtype=F.dtype)
    dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM)
    return out[:row1 - row0], d[row0:row1], row0


def gather_rows(rows: torch.Tensor, n: int) -> torch.Tensor:
    """All ranks' finalised row blocks -> the whole [n, D] field on every rank (one all-gather; the reference writes one
    features file, backproject.py:330)."""
    dist, _, world = _dist()
    if dist is None:
        return rows
    per = rows_per_rank(n, world)
    if rows.shape[0] != per:  # the last rank's short block: pad to the common block size
        rows = torch.cat([rows, rows.new_zeros(per - rows.shape[0], rows.shape[1])])
    full = torch.empty(per * world, rows.shape[1], device=rows.device, dtype=rows.dtype)
    dist.all_gather_into_tensor(full, rows.contiguous())
    return full[:n]


def finalize_reference(F: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """backproject.py:63,166-169 in plain torch (host logic used by the CPU/gloo tests; the GPU path calls
    Engine.finalize -> gwbp_finalize)."""
    den = 1e-12 + d
    x = F / den[:, None]
    x = x / x.norm(dim=-1, keepdim=True)
    x[torch.isnan(x)] = 0
    return x


# verdict of wide_kernel_selfcheck per (library path, device index): the check runs once per process and device
_WIDE_OK: Dict[tuple, bool] = {}
WIDE_SELFCHECK_TOL = 1e-6


def wide_kernel_selfcheck(device) -> bool:
    """Does the 256-channel scatter kernel of THIS build agree with the 128-channel one on THIS device?

    k_scatter_wide keeps asm-issued loads and two fixed SGPR tuples live across inline-asm statements; the build is gated by an
    assembly scan (tools/check_asm_hazards.py, run by the Makefile for every variant), and this is the run-time half of the same
    safety net: one small seeded view (T1: 4000 Gaussians, 200x136, D = 256 -- records in one half, in both halves, padded
    lists, carry rows) is blended once and scattered through both kernels; they must agree to 1e-6 of the largest row (the
    kernels differ only in summation order: 1.4e-7 measured over every build variant of round 5).  On disagreement
    ViewPipeline.choose_scatter_kernel warns and stays on the 128-channel kernel -- 5 % slower at BASELINE size, never wrong.
    A few milliseconds, once per process and device, before the first D % 256 == 0 job."""
    from . import _lib, synthetic as syn
    dev = torch.device(device)
    key = (_lib._lib_path, dev.index if dev.index is not None else torch.cuda.current_device())
    if key in _WIDE_OK:
        return _WIDE_OK[key]
    cfg = syn.CONFIGS["T1"]
    means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
    K, vm = syn.intrinsics(cfg), syn.make_cameras(cfg)[0]
    feats = syn.make_feature_map(cfg, 0, device=dev, dim=256)
    res = []
    for wide in (True, False):
        eng = Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, tight_binning=True)
        eng.set_narrow_scatter(not wide)
        view = eng.view(vm, K, cfg.width, cfg.height)
        eng.project(view, means, quats, scales, opac)
        eng.bin_sort(view)
        eng.blend_weights(view)
        F = torch.zeros(cfg.n_gaussians, 256, device=dev)
        eng.scatter(view, feats, F, None)
        res.append(F)
    scale = res[1].norm(dim=1).max().clamp_min(1e-30)
    diff = (res[0] - res[1]).norm(dim=1).max() / scale
    ok = bool(torch.isfinite(res[0]).all()) and bool(res[1].abs().sum() > 0) and float(diff) <= WIDE_SELFCHECK_TOL
    _WIDE_OK[key] = ok
    return ok


class ViewPipeline:
    """Software pipeline over views on the caller's stream + side streams, one workspace per view in flight (two by default).

    front(v)   = project -> bin/sort -> blend_weights   (small latency-bound kernels + the 20 KB-LDS blend)
    scatter(v) = the weighted scatter-accumulate         (one 139 KB-LDS workgroup per CU: 16 of 32 wave slots)
    front(v+1) runs on a side stream while scatter(v) runs on the caller's stream, so the front kernels fill
    the wave slots the scatter kernel cannot use.  Events order  front(v) -> scatter(v) -> front(v+K)
    (workspace reuse, K = number of engines).  Nothing synchronises the host.

    With K > 2 engines the fronts of views v+1 .. v+K-1 run concurrently on K-1 side streams (`lookahead` = K - 1): on small
    scenes the front stage is a chain of ~30 dependent launches that does not fill the chip, and several chains in flight
    multiply its throughput (pipeline_depth()).
    """

    def __init__(self, n_gaussians, width, height, device, engines=None, scatter_dim: Optional[int] = None,
                 allow_wide: bool = True, scatter_workgroups: Optional[int] = None, side_priority: int = -1,
                 front_priority: Optional[bool] = None, fuse_small: bool = True, side_streams: Optional[int] = None,
                 view_per_stream: Optional[bool] = None, token_grid=None, split_encoder: bool = False):
        self.dev = torch.device(device)
        # token_grid = (h, w): the views' feature maps are h x w maps upsampled with mode="nearest" whose texels are at least a
        # tile wide and high (the dino variant's 64 x 64 patch tokens): the front stage ends with Engine.blend_tokens (per-record
        # token-quadrant weight sums instead of a weight store) and scatter() runs Engine.scatter_tokens -- one plain
        # read-modify-write of every F row that receives weight, no atomics (csrc/token.hip)
        self.token_grid = tuple(int(x) for x in token_grid) if token_grid is not None else None
        from . import _lib
        if _lib.hw_queues_late():
            # the schedule below is built on streams that run CONCURRENTLY; with the HIP runtime's default of 4 hardware queues
            # (fewer once RCCL has taken its own) the side streams share a queue with the caller's and the step becomes front +
            # scatter.  The package asks for 8 when it is imported (_lib.py), which only works before the first HIP call.
            raise _lib.GwbpError(
                "ViewPipeline needs GPU_MAX_HW_QUEUES >= %d, read by the HIP runtime when it starts: import gsbp_amd (or set "
                "the variable) before the first CUDA/HIP call of the process, or run the views with pipeline=False"
                % _lib.HW_QUEUES_WANTED)
        if not _lib.hw_queues_ok():  # somebody CHOSE fewer queues (the variable was set before the package was imported)
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES=%s < %d: the view pipeline's streams may share hardware queues and run back to back"
                          % (__import__("os").environ.get("GPU_MAX_HW_QUEUES"), _lib.HW_QUEUES_WANTED), RuntimeWarning)
        self.eng = list(engines) if engines else [Engine(n_gaussians, width, height, device=self.dev, tight_binning=True)
                                                  for _ in range(2)]
        # Scatter grid under overlap: one persistent workgroup per CU is the measured optimum once the front stage is
        # light (C2: 4.24 ms/view at 256 vs 4.40 at 240); `scatter_workgroups` overrides.
        if scatter_workgroups is not None:  # tuning on other workloads
            for e in self.eng:
                e.scatter_workgroups = int(scatter_workgroups)
                e.caps.scatter_workgroups = e.scatter_workgroups
        # High priority = a hardware queue of its own.  With default priority the side stream can land on the main
        # stream's hardware queue (it does once RCCL has created its streams: GPU_MAX_HW_QUEUES is 4), the two streams
        # then run strictly one after the other and the step is front + scatter (5.06 instead of 4.27 ms/view at C2).
        self.scatter_dim = scatter_dim
        self.allow_wide = bool(allow_wide)
        # Narrow maps (D <= 16): the front stage stops after the sort, and blend + scatter run as ONE kernel on the caller's
        # stream (gwbp_blend_scatter: no weight store, no scatter kernel); the side stream keeps project + sort of view v+1.
        # On small images the kernel runs a wave per QUARTER tile (four short blend chains per tile instead of one long
        # one) and takes up to 32 channels; on large ones it needs two workspaces' worth of schedule (len(eng) == 2).
        small_image = Engine.fused_max_dim(width, height) == Engine.FUSED_MAX_DIM_SMALL
        # view_per_stream=True: large images too run every view entirely on a stream of its own (K > 2 workspaces): the fused
        # kernels of consecutive views then overlap at their edges (measured for the compressed variant)
        self.fuse_small = (bool(fuse_small) and scatter_dim is not None and self.token_grid is None
                           and scatter_dim <= Engine.fused_max_dim(width, height)
                           and (small_image or len(self.eng) == 2 or bool(view_per_stream)))
        # split_encoder: blend_scatter_encoded in its producer / consumer form (GWBP_FLAG_SPLIT_ENCODER: one persistent launch per
        # view whose encoder waves and blend waves run concurrently; the compressed variant on large images)
        self.split_encoder = bool(split_encoder)
        for e in self.eng:
            e.set_split_encoder(self.split_encoder)
        self.front_priority = front_priority  # None: raised wave priority for the front exactly when the wide kernel runs
        self.choose_scatter_kernel(None, None)
        K = len(self.eng)
        self.lookahead = K - 1  # fronts the driver keeps enqueued ahead of the scatter
        # Small scenes whose blend + scatter is the fused kernel (K > 2 workspaces): every workspace gets a stream of its own
        # and a view runs ENTIRELY on it -- project, sort, blend+scatter in stream order, K views in flight, one event per
        # view (the map must be ready) instead of two, and the fused kernels of consecutive views overlap.  F and d take
        # atomics anyway.  (With separate blend and scatter kernels this schedule was measured to change nothing.)
        self.independent = K > 2 and self.fuse_small and view_per_stream is not False
        n_side = K if self.independent else max(1, K - 1)
        if side_streams is not None and not self.independent:
            n_side = max(1, min(int(side_streams), K - 1))
        self.sides = [torch.cuda.Stream(device=self.dev, priority=int(side_priority)) for _ in range(n_side)]
        self.side = self.sides[0]
        self.enc_stream = None  # encoder stream, created by the first encode_ahead()
        self.ev_front = [torch.cuda.Event() for _ in range(K)]
        self.ev_done = [torch.cuda.Event() for _ in range(K)]
        # counters: one accumulator per stream that adds to it (k_accum_stats is a plain read-modify-write)
        self.accums = [torch.zeros(32, dtype=torch.uint8, device=self.dev) for _ in range(K if self.independent else 1)]
        self.accum = self.accums[0]
        if self.independent:  # the host is the limit on such scenes: no stream context switch per call either
            for e, st in zip(self.eng, self.sides):
                e.bind_stream(st)
            self.ev_ready = [torch.cuda.Event() for _ in range(K)]
        self.ev_slot = [torch.cuda.Event() for _ in range(K)]  # view-per-stream schedule: "the view in this slot is done"
        self.slot_used = [False] * K
        self.i_front = 0    # views whose front stage has been enqueued
        self.i_scatter = 0  # views whose scatter stage has been enqueued
        self.pending = {}

    # records shorter than this many pairs on average: the 128-channel scatter kernel (C2 48 -> wide, C4 25 -> narrow)
    WIDE_MIN_PAIRS_PER_RECORD = 36.0

    def choose_scatter_kernel(self, n_pairs, n_headers) -> str:
        """D % 256 == 0 may go through the 256-channel scatter kernel.  It needs fewer vector instructions per (pair,
        channel) and leaves issue slots that the front stage can use at raised wave priority (4.08 vs 4.30 ms/view at
        C2), but costs more per (Gaussian, tile) record: with short records (C4: 25 pairs per record) it loses 5 %, and
        beside the 128-channel kernel the raised priority costs 6 %.  Called once without statistics (wide if the channel
        count allows) and again by the drivers with the first views' counters.  Safe in either direction while fronts
        are pending: every Engine remembers whether the view in its workspace was blended with the half-tile lists, and
        scatters a view blended without them through the 128-channel kernel (front() likewise recorded whether it has
        already added that view's denominators)."""
        wide = (self.allow_wide and self.scatter_dim is not None and self.scatter_dim % 256 == 0
                and self.token_grid is None)  # (token space: neither scatter kernel runs)
        if wide and n_pairs is not None and n_headers:
            wide = n_pairs / n_headers >= self.WIDE_MIN_PAIRS_PER_RECORD
        if wide and self.dev.type == "cuda" and not wide_kernel_selfcheck(self.dev):
            import warnings
            warnings.warn("the 256-channel scatter kernel of this build disagrees with the 128-channel one on the self-check "
                          "view (wide_kernel_selfcheck): using the 128-channel kernel.  Rebuild libgwbp.so "
                          "(python __graft_entry__.py) and report the toolchain version.", RuntimeWarning)
            wide = False
        self.wide = wide
        for e in self.eng:
            e.set_narrow_scatter(not wide)
            e.set_front_priority(wide if self.front_priority is None else bool(self.front_priority))
        return "wide" if wide else "narrow"

    def front(self, view, means, quats, scales, opacities, d=None, scale_d=1.0):
        """d (optional): the denominator accumulator.  With the 256-channel scatter kernel chosen, the view's share of d
        is added by the blend itself on the side stream (gwbp_blend_weights_d) and scatter() then leaves d alone: the
        denominators cost nothing on the scatter's stream."""
        K = len(self.eng)
        b = self.i_front % K
        side = self.sides[self.i_front % len(self.sides)]
        main = torch.cuda.current_stream(self.dev)
        if self.i_front < K:
            side.wait_stream(main)  # inputs produced on the caller's stream
        elif not self.independent:
            side.wait_event(self.ev_done[b])  # workspace b is free once scatter(i-K) has finished
        # (independent: scatter(i-K) was enqueued on this very stream)
        if self.independent:  # the engine is bound to `side`
            e = self.eng[b]
            e.project(view, means, quats, scales, opacities)
            e.bin_sort(view)
            self.pending[self.i_front] = (view, False, False, False)
            self.i_front += 1
            return
        with torch.cuda.stream(side):
            e = self.eng[b]
            e.project(view, means, quats, scales, opacities)
            e.bin_sort(view)
            d_done = d is not None and self.wide and not self.fuse_small and self.token_grid is None
            if self.token_grid is not None:
                e.blend_tokens(view, *self.token_grid)
            elif not self.fuse_small:
                e.blend_weights(view, d=d if d_done else None, scale_d=scale_d)
            self.ev_front[b].record(side)
        self.pending[self.i_front] = (view, d_done, not self.fuse_small, self.token_grid is not None)
        self.i_front += 1

    # Encoder workgroups per CU beside the pipeline (see encode_ahead): None = 1 next to the separate blend and small-D
    # scatter kernels (both latency-bound: C5 2.19 -> 1.96 ms/view against 4 per CU), 2 next to the fused blend+scatter
    # kernel, where one per CU makes the encoder itself the long pole (C5: 1.65 ms/view at 1, 1.42 at 2, 1.45 at 3)
    ENCODER_WORKGROUPS_PER_CU = None

    def encode_ahead(self, feats: torch.Tensor, encoder: torch.Tensor):
        """The compressed variant's per-pixel encoder (backproject_compressed.py:127) for a LATER view on a third stream:
        an HBM-streaming kernel that overlaps with the latency-bound scatter of the current view and the front stage of
        the next.  Call it for view v+1 before scatter(v) is enqueued; hand the returned (map, event) to scatter()."""
        main = torch.cuda.current_stream(self.dev)
        if self.enc_stream is None:
            self.enc_stream = torch.cuda.Stream(device=self.dev, priority=-1)
        ready = torch.cuda.Event()
        ready.record(main)  # the map was produced on the caller's stream
        with torch.cuda.stream(self.enc_stream):
            self.enc_stream.wait_event(ready)
            # one workgroup per CU: streaming harder doubles the memory latency of the front stage and the scatter beside it
            n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
            per_cu = self.ENCODER_WORKGROUPS_PER_CU or (2.0 if self.fuse_small else 1.0)
            # stream=: in the view-per-stream schedule eng[0] is BOUND to sides[0]; the encoder must run here, behind
            # `ready`, and `done` must cover it
            out = encode_features(self.eng[0], feats, encoder, workgroups=max(1, int(per_cu * n_cu)), stream=self.enc_stream)
            done = torch.cuda.Event()
            done.record(self.enc_stream)
        feats.record_stream(self.enc_stream)
        out.record_stream(main)
        return out, done

    def scatter(self, feats, F, d, scale_f=1.0, scale_d=1.0, t0=None, t1=None, upsample=None, after=None, encoder=None,
                ready: bool = False):
        """t0/t1: optional timing events recorded right around the scatter launch (after the cross-stream waits).
        after: an event the feature map depends on (encode_ahead).
        encoder: scatter feats @ encoder with the encoder fused into the slab staging (Engine.scatter_encoded).
        ready: the map needs no cross-stream wait -- it was produced on scatter_stream() (a feature function run under
        `with torch.cuda.stream(pipe.scatter_stream())`) or is known to be complete (a pool built before the job); only
        looked at by the view-per-stream schedule, where that wait is a third of the host's time per view."""
        i = self.i_scatter
        b = i % len(self.eng)
        main = torch.cuda.current_stream(self.dev)
        if self.independent:
            # on the view's own stream, behind its front; the map was produced on the caller's stream
            side = self.sides[b]
            if not ready:
                ev = self.ev_ready[b]  # (its previous use, view i - K, was waited for on this same stream long ago)
                ev.record(main)
                side.wait_event(ev)
                feats.record_stream(side)
            if after is not None:
                side.wait_event(after)
            self._scatter_on(side, b, feats, F, d, scale_f, scale_d, t0, t1, upsample, encoder)  # engine bound to `side`
            self.ev_slot[b].record(side)
            self.slot_used[b] = True
            self.i_scatter += 1
            return
        if after is not None:
            main.wait_event(after)
        main.wait_event(self.ev_front[b])
        self._scatter_on(main, b, feats, F, d, scale_f, scale_d, t0, t1, upsample, encoder)
        self.ev_done[b].record(main)
        self.i_scatter += 1

    def _scatter_on(self, main, b, feats, F, d, scale_f, scale_d, t0, t1, upsample, encoder):
        i = self.i_scatter
        e = self.eng[b]
        if t0 is not None:
            t0.record(main)
        view, d_done, blended, tok = self.pending.pop(i)
        fused = not blended and encoder is None and upsample is None and Engine.can_blend_scatter(feats)
        if tok:
            if (upsample == "nearest" and encoder is None and tuple(feats.shape[:2]) == self.token_grid
                    and Engine.can_scatter_tokens(feats, view.height, view.width)):
                e.scatter_tokens(view, feats, F, d, scale_f, scale_d)
            else:  # this view's map does not suit the token path after all: blend a weight store here, then the usual scatter
                e.blend_weights(view)
                if encoder is not None:
                    e.scatter_encoded(view, feats, encoder, F, d, scale_f, scale_d)
                else:
                    e.scatter(view, feats, F, d, scale_f, scale_d, upsample=upsample)
        elif fused:
            e.blend_scatter(view, feats, F, d, scale_f, scale_d)
        elif not blended and encoder is not None and upsample is None and Engine.can_blend_scatter_encoded(feats, encoder):
            # the compressed variant in ONE kernel: encoder in the tile prologue, then blend + scatter from registers
            e.blend_scatter_encoded(view, feats, encoder, F, d, scale_f, scale_d)
        else:
            if not blended:  # the map does not suit the fused kernel after all: blend here, then the usual scatter
                e.blend_weights(view)
            if encoder is not None:
                e.scatter_encoded(view, feats, encoder, F, None if d_done else d, scale_f, scale_d)
            else:
                e.scatter(view, feats, F, None if d_done else d, scale_f, scale_d, upsample=upsample)
        if t1 is not None:
            t1.record(main)
        e.accumulate_stats(self.accums[b if self.independent else 0])

    def wait_for_slot(self) -> None:
        """View-per-stream schedule: block the HOST until the view that last used the next scatter's slot is done, i.e. keep the
        host at most K views ahead of the device.  Call it before producing a view's feature map.  Without it nothing
        throttles the host there (a view's kernels wait for nothing the host waits for), and a feature function that
        allocates a fresh map per view -- 3.47 GB at C2 size -- runs the caching allocator into the ground: a map handed to a
        side stream is only recycled once that stream has passed it, so the host allocates dozens of maps before the first is
        free again, and the allocator's out-of-memory path (synchronise, release, retry) then stalls the job for seconds
        (the CLI at C5 size: 1.1 s or 6 s for the same command).  No-op in the other schedules: there the maps are
        consumed on the caller's stream and recycled in stream order."""
        if self.independent:
            b = self.i_scatter % len(self.eng)
            if self.slot_used[b]:
                self.ev_slot[b].synchronize()

    def scatter_stream(self) -> torch.cuda.Stream:
        """The stream the NEXT scatter() runs on (the caller's current stream unless a view runs on a stream of its own)."""
        if self.independent:
            return self.sides[self.i_scatter % len(self.eng)]
        return torch.cuda.current_stream(self.dev)

    def stream_of(self, i: int) -> torch.cuda.Stream:
        """The stream view i's FRONT stage runs on (timing events of a driver)."""
        return self.sides[i % len(self.sides)]

    def join(self):
        """Make the caller's stream wait for everything enqueued so far (the accumulators are complete behind it)."""
        main = torch.cuda.current_stream(self.dev)
        for side in self.sides:
            main.wait_stream(side)

    def release(self):
        """End of the job: the caller's stream waits for everything enqueued, and engines that were bound to a stream of
        this pipeline follow torch's current stream again (an engine handed in by the caller outlives the pipeline)."""
        self.join()
        if self.independent:
            for e in self.eng:
                e.bind_stream(None)

    def reset_stats(self):
        self.join()
        for a in self.accums:
            a.zero_()

    def stats_async(self):
        """(host copies of the counter accumulators, their events): enqueue a device-to-pinned-host copy of the counters behind
        the work enqueued so far on their streams; when every event.query() is true the copies may be decoded
        (Engine.decode_stats).  Nothing synchronises."""
        copies, events = [], []
        streams = self.sides if self.independent else [torch.cuda.current_stream(self.dev)]
        for a, st in zip(self.accums, streams):
            host = torch.empty(a.shape, dtype=a.dtype, pin_memory=True)
            with torch.cuda.stream(st):
                host.copy_(a, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            copies.append(host)
            events.append(ev)
        return copies, events

    def stats(self):
        """Counters summed over the views scattered so far; synchronises."""
        self.join()
        out: Dict[str, int] = {}
        for a in self.accums:
            for k, v in Engine.decode_stats(a).items():
                if k == "blend_kind":  # a per-view fact, not a counter
                    continue
                out[k] = (out.get(k, 0) | v) if k == "overflow" else out.get(k, 0) + v
        return out


# images of at least this many 16 x 16 tiles run the compressed variant's fused kernel in its producer / consumer form by default
# (create_feature_field(encoder_split=None)): 16 tiles per CU on MI355X, below which the ring's start-up and tail outweigh the overlap
SPLIT_ENCODER_MIN_TILES = 4096

# create_feature_field looks at the accumulated overflow flags after view 2 and then every this many views (a non-blocking
# copy of the counters, read back a few views later): a capacity overflow in view 150 of 200 restarts the job after at most
# this many more views instead of after the last one
OVERFLOW_CHECK_EVERY = 32


def pipeline_depth(n_gaussians: int, width: int, height: int, dim: Optional[int] = None, encoder_in_blend: bool = False) -> int:
    """Workspaces (views in flight) of the ViewPipeline.
    Small scenes: 4 -- three front stages in flight on three side streams: a view is ~35 dependent launches of 5-20 us each
    and the chain, not the chip, is the limit (C1: 0.49 -> 0.19 ms/view; 5 and more lose again: the streams start sharing
    hardware queues).
    Large scenes: 3.  Since round 4 the 256-channel scatter kernel prefetches its next slab under the tail of the running
    pass, which leaves the ONE front stage beside it too few idle slots to keep up (C2 at depth 2: front 3.82 ms against a
    3.70 ms scatter); with two fronts in flight the scatter kernel is the long stage again (3.91 -> 3.71 ms/view; C4
    10.15 -> 10.02; a fourth workspace adds nothing).  Maps narrow enough for the fused blend + scatter kernel (`dim` <= 16 on
    large images, the compressed variant) keep 2: that kernel's schedule is built for two workspaces (C5: 1.47 against 1.98).
    `encoder_in_blend` (round 5: the compressed variant's encoder inside the fused kernel's tile prologue,
    gwbp_blend_scatter_encoded): 4, every view entirely on a stream of its own -- one such kernel runs its HBM-bound prologues
    and its issue-bound blend loops in lockstep (1.29 ms alone = encoder 0.58 + blend 0.70), the kernels of four views in
    flight overlap them (C5: 1.27-1.33 ms/view; 3 workspaces 1.37-1.41, 5: 1.39-1.41, 6: 1.36-1.37; the separate encoder one
    view ahead: 1.38-1.44)."""
    if n_gaussians <= 250_000 and width * height <= 1_000_000:
        return 4
    if encoder_in_blend:
        return 4
    if dim is not None and dim <= Engine.fused_max_dim(width, height):
        return 2
    return 3


def create_feature_field(means, quats, scales, opacities, viewmats, K, width: int, height: int,
                         feature_fn: Callable[[int], torch.Tensor], dim: int, reduction: str = "sum",
                         encoder: Optional[torch.Tensor] = None, engine: Optional[Engine] = None,
                         views: Optional[Sequence[int]] = None, view_fn=None, pipeline: bool = True,
                         return_partials: bool = False, verbose: bool = False, upsample: Optional[str] = None,
                         gather: bool = True, allow_wide: bool = True, fuse_encoder: bool = False,
                         fuse_small: bool = True, feature_fn_stream_safe: bool = False,
                         encoder_in_blend: Optional[bool] = None, token_space: bool = True,
                         encoder_split: Optional[bool] = None):
    """Build the [N, dim_out] per-Gaussian feature field.

    means/quats/scales/opacities: post-activation Gaussians (backproject.py:55-57), device tensors.
    viewmats [V,4,4], K [3,3]; feature_fn(v) -> feats[H,W,dim] float32 on the same device (stands in for the
    LSeg/DINO forward of backproject.py:102-113 / :236-249).
    reduction: "sum" (lseg, backproject.py:127,145) or "mean" (dino, backproject.py:263,283).
    encoder [dim, dim_out]: backproject_compressed.py:127 (feats @ encoder before back-projection).
    upsample="nearest" | "bilinear": feature_fn returns the network's LOW-RESOLUTION map [h,w,dim] (dino patch tokens,
    backproject.py:242-243; the normalised lseg map, :108-109); the upsampling to (height, width) of
    backproject.py:244-248 / :110-112 happens while the scatter kernel stages its tile slabs instead of materialising
    an [H,W,dim] map per view.
    views: explicit list of view indices for this rank (default: interleaved shard over the process group).
    view_fn: injection point for the per-view accumulate (tests drive the sharding/reduction logic on CPU).
    fuse_encoder: with `encoder`, apply it inside the scatter kernel's slab staging (gwbp_scatter_encoded: the full-width map
    is read once, no [H,W,dim_out] intermediate) when the map's layout allows; False (default, measured faster in the
    three-stream pipeline: C5 2.19 vs 2.29 ms/view; the fused kernel wins on one stream, 2.51 vs 2.70) = a separate encode
    kernel one view ahead on a third stream (gwbp_encode_map).
    encoder_in_blend: with `encoder`, apply it inside the fused blend + scatter kernel's tile prologue (gwbp_blend_scatter_encoded:
    one kernel per view, no [H,W,dim_out] map, four views in flight on streams of their own).  None (default) = whenever the
    first map's layout allows and fuse_small is on (C5: 1.27-1.33 against 1.38-1.44 ms/view for the encoder one view ahead);
    False = never.
    encoder_split: with encoder_in_blend, run that kernel in its producer / consumer form (GWBP_FLAG_SPLIT_ENCODER: one persistent
    launch per view, encoder waves and blend waves around an LDS ring of encoded tiles).  None (default) = on images of at least
    SPLIT_ENCODER_MIN_TILES tiles, where a CU's encoder waves have tiles enough to stream; False = never.
    token_space: with upsample="nearest", maps whose texels are at least a 16 x 16 tile wide and high and whose channel count is a
    multiple of 4, 64 or more (the dino variant's 64 x 64 x 1024 patch tokens at 1600 x 1060; the other DINOv2 backbones' 384,
    768, 1536) are back-projected in TOKEN space: the blend
    leaves per-(Gaussian, tile) weight sums of the tile's 2 x 2 tokens and every F row that receives weight is updated with ONE
    plain read-modify-write per view from the L2-resident token map -- no atomics, no weight store (csrc/token.hip).  False keeps
    the pixel-slab kernels (gwbp_scatter_upsampled) for such maps too.
    fuse_small: maps of at most 16 channels (after the encoder) are blended AND scattered by one kernel
    (gwbp_blend_scatter: no weight store, no scatter kernel; C5 1.96 -> 1.42 ms/view); False keeps the two-kernel form.
    feature_fn_stream_safe: STREAM CONTRACT of feature_fn.  False (default): feature_fn runs on the caller's current stream
    and every map is handed to its consumer stream with an event -- any feature function is safe, including one that
    returns a prefetched tensor, reuses a static output buffer or replays a graph (the buffer must still not be overwritten
    before the view that reads it has been scattered: `pipeline` views may be in flight).  True: in the view-per-stream
    schedule (small scenes, <= 32-channel maps) feature_fn is called under `torch.cuda.stream(<the consuming stream>)` and
    no event is recorded (a third of the host time per view there); only for functions that allocate and produce their
    output entirely on torch's current stream at call time.
    pipeline: overlap the front stages of the next view(s) with the scatter of view v (ViewPipeline); True = depth chosen by
    pipeline_depth(N, width, height, dim_out), an int >= 2 = that many workspaces, False = one stream.
    gather: under a process group, all-gather the finalised row blocks so that every rank returns the whole [N, dim_out]
    field; False returns this rank's block only (rows row0 .. of `return_partials`' stats["row0"]).
    return_partials: also return (F_rows, d, stats): the summed, un-normalised accumulators (this rank's row block of F,
    all of d) and the counters.
    """
    dist, rank, world = _dist()
    n = means.shape[0]
    d_out = dim if encoder is None else encoder.shape[1]
    dev = means.device
    F, d, F_store = alloc_accumulators(n, d_out, dev, world)
    my_views = list(views) if views is not None else view_shard(viewmats.shape[0], rank, world)
    vm_host, K_host = viewmats.detach().cpu(), K.detach().cpu()  # one D2H copy, not one per view
    if reduction == "sum":
        sf, sd = 1.0, 1.0
    elif reduction == "mean":
        sf, sd = 1.0 / (height * width * d_out), 1.0 / (height * width * 3)
    else:
        raise ValueError(reduction)

    t0 = time.time()
    stats: Dict[str, int] = {}
    if view_fn is None:
        eng = engine or Engine(n, width, height, device=dev, tight_binning=True)  # same F and d, shorter tile lists
        if pipeline and len(my_views) > 1:
            from . import _lib
            if _lib.hw_queues_late():
                # Somebody touched the GPU before this package was imported (a notebook, another library): the request for
                # hardware queues came too late and the pipeline's streams would share queues.  That costs speed, not results,
                # so the DRIVER degrades by itself: one stream, same F and d.  Constructing a ViewPipeline
                # explicitly still raises -- whoever asks for the overlapped schedule by name should hear that it cannot run.
                import warnings
                warnings.warn("gsbp_amd was imported after the HIP runtime had started: GPU_MAX_HW_QUEUES=%d could not be "
                              "requested, create_feature_field runs its views on ONE stream (pipeline=False; ~20 %% slower at "
                              "BASELINE size).  Import gsbp_amd, or set the variable, before the first CUDA/HIP call."
                              % _lib.HW_QUEUES_WANTED, RuntimeWarning, stacklevel=2)
                pipeline = False
        split_allowed = True
        for attempt in range(6):  # a capacity overflow invalidates the accumulators: grow the workspace, start over
            if pipeline and len(my_views) > 1:
                first_map = feature_fn(my_views[0]) if (encoder is not None or upsample == "nearest") else None
                # the dino shape -- a nearest-upsampled map whose texels are at least a tile wide and high, D % 4 == 0 -- goes
                # through token space (Engine.blend_tokens / scatter_tokens); decided on the first map, checked per view
                token_grid = (tuple(first_map.shape[:2]) if (token_space and upsample == "nearest" and encoder is None
                                                             and Engine.can_scatter_tokens(first_map, height, width)) else None)
                enc_blend = (encoder is not None and fuse_small and not fuse_encoder and upsample is None
                             and encoder_in_blend is not False and Engine.can_blend_scatter_encoded(first_map, encoder))
                depth = (pipeline_depth(n, width, height, d_out, encoder_in_blend=enc_blend) if pipeline is True
                         else max(2, int(pipeline)))
                split = bool(enc_blend and split_allowed and (encoder_split if encoder_split is not None
                                            else (-(-width // 16)) * (-(-height // 16)) >= SPLIT_ENCODER_MIN_TILES))
                pipe = ViewPipeline(n, width, height, dev, scatter_dim=d_out, token_grid=token_grid, split_encoder=split,
                                    allow_wide=allow_wide, fuse_small=fuse_small and not (fuse_encoder and encoder is not None),
                                    view_per_stream=True if (enc_blend and depth > 2) else None,
                                    engines=[eng] + [Engine(n, width, height, device=dev, tight_binning=eng.tight_binning,
                                                            isect_cap=eng.isect_cap, pair_cap=eng.pair_cap)
                                                     for _ in range(depth - 1)])
                views = [eng.view(vm_host[v], K_host, width, height) for v in my_views]
                for j in range(min(pipe.lookahead, len(my_views))):
                    pipe.front(views[j], means, quats, scales, opacities, d, sd)
                # with an encoder the feature function runs one view ahead, so that view v+1's map is encoded on a third
                # stream while view v is scattered (two full-width maps are alive at a time)
                fused = None  # decided on the first map: its layout must suit gwbp_scatter_encoded
                ahead = None
                if encoder is not None and not enc_blend:
                    fused = fuse_encoder and upsample is None and Engine.can_fuse_encoder(first_map, encoder)
                    ahead = (first_map, None) if fused else pipe.encode_ahead(first_map, encoder)
                probe = None  # (pinned copy of the counters, event): an overflow costs at most OVERFLOW_CHECK_EVERY views
                overflowed = False
                for i, v in enumerate(my_views):
                    if i == 2:  # one host sync per job: views 0 and 1 are counted, pick the scatter kernel for the rest
                        st01 = pipe.stats()
                        if st01["overflow"]:
                            overflowed = True
                            break
                        pipe.choose_scatter_kernel(st01["n_pairs"], st01["n_headers"])
                    if probe is not None and all(e.query() for e in probe[1]):  # non-blocking: the copy was enqueued views ago
                        if any(Engine.decode_stats(a)["overflow"] for a in probe[0]):
                            overflowed = True
                            break
                        probe = None
                    if i > 2 and i % OVERFLOW_CHECK_EVERY == 0 and probe is None:
                        probe = pipe.stats_async()
                    if i + pipe.lookahead < len(my_views):
                        pipe.front(views[i + pipe.lookahead], means, quats, scales, opacities, d, sd)
                    pipe.wait_for_slot()  # (view-per-stream schedule only: the host stays at most `depth` maps ahead)
                    if enc_blend:
                        # encoder + blend + scatter of the view in one kernel, on the view's own stream
                        feats = first_map if i == 0 else feature_fn(v)
                        pipe.scatter(feats, F, d, sf, sd, encoder=encoder)
                        continue
                    if fused:
                        # the encoder is applied inside the scatter kernel's slab staging: no [H,W,dim_out] map at all
                        feats = ahead[0] if i == 0 else feature_fn(v)
                        if not Engine.can_fuse_encoder(feats, encoder):
                            feats, fenc = encode_features(eng, feats, encoder), None
                        else:
                            fenc = encoder
                        pipe.scatter(feats, F, d, sf, sd, encoder=fenc)
                        continue
                    if encoder is not None:
                        feats, after = ahead
                        if i + 1 < len(my_views):
                            ahead = pipe.encode_ahead(feature_fn(my_views[i + 1]), encoder)
                    elif feature_fn_stream_safe:
                        # (view-per-stream schedule: the feature function runs on the stream that consumes its map)
                        with torch.cuda.stream(pipe.scatter_stream()):
                            feats, after = feature_fn(v), None
                    elif upsample == "nearest" and i == 0:
                        feats, after = first_map, None  # (already produced to read its shape)
                    else:
                        feats, after = feature_fn(v), None  # on the caller's stream; scatter() waits for it with an event
                    pipe.scatter(feats, F, d, sf, sd, upsample=upsample, after=after,
                                 ready=encoder is None and feature_fn_stream_safe)
                stats = pipe.stats()
                pipe.release()
                assert not overflowed or stats["overflow"]
            else:
                accum = torch.zeros(32, dtype=torch.uint8, device=dev)
                for i, v in enumerate(my_views):
                    feats = feature_fn(v)
                    if encoder is not None:
                        feats = encode_features(eng, feats, encoder)
                    view = eng.view(vm_host[v], K_host, width, height)
                    if upsample is None and fuse_small and Engine.can_blend_scatter(feats):
                        eng.project(view, means, quats, scales, opacities)
                        eng.bin_sort(view)
                        eng.blend_scatter(view, feats, F, d, sf, sd)
                    elif upsample is None:
                        eng.backproject_view(view, means, quats, scales, opacities, feats, F, d, sf, sd)
                    elif token_space and upsample == "nearest" and Engine.can_scatter_tokens(feats, height, width):
                        eng.project(view, means, quats, scales, opacities)
                        eng.bin_sort(view)
                        eng.blend_tokens(view, feats.shape[0], feats.shape[1])
                        eng.scatter_tokens(view, feats, F, d, sf, sd)
                    else:
                        eng.project(view, means, quats, scales, opacities)
                        eng.bin_sort(view)
                        eng.blend_weights(view)
                        eng.scatter(view, feats, F, d, sf, sd, upsample=upsample)
                    eng.accumulate_stats(accum)
                stats = Engine.decode_stats(accum)  # synchronises
            if stats["overflow"] & 16:
                # a wave of the producer / consumer kernel gave up waiting on its LDS ring (its waits are bounded so that a
                # scheduling accident cannot hang the device): the accumulators are incomplete.  Never seen in 2000-view soaks
                # and the fuzz; if it happens, say so and build the field again with the one-wave-per-tile form of the kernel.
                if not split_allowed:
                    raise RuntimeError("gwbp_stats.overflow bit 4 (ring stall) without the producer / consumer kernel: internal error")
                import warnings
                warnings.warn("a wave of gwbp_blend_scatter_encoded's producer / consumer kernel gave up waiting on its LDS ring "
                              "(gwbp_stats.overflow bit 4); building the field again with encoder_split=False -- please report it",
                              RuntimeWarning, stacklevel=2)
                split_allowed = False
                F.zero_()
                d.zero_()
                continue
            if stats["overflow"] & 8:
                raise RuntimeError("gwbp_blend_tokens met a tile that spans more than 2 x 2 texels (gwbp_stats.overflow bit 3): "
                                   "the map is finer than Engine.token_geometry_ok() admitted")
            if stats["overflow"] & 4:
                raise RuntimeError("a view was scattered with a kernel that did not match its blend (gwbp_stats.overflow "
                                   "bit 2): F and d are incomplete")
            if not stats["overflow"]:
                break
            if attempt == 5:
                raise RuntimeError(f"workspace overflow (flags {stats['overflow']}) after five enlargements")
            eng.grow(stats, views=len(my_views))
            F.zero_()
            d.zero_()
    else:
        eng = None
        for v in my_views:
            feats = feature_fn(v)
            if encoder is not None:
                feats = feats @ encoder
            view_fn(v, feats)
    # The single exchange step: reduce-scatter of F (+ all-reduce of d), row-local finalise of this rank's block
    # (backproject.py:166-169 needs nothing but the row), and an all-gather of the finalised blocks only when the caller
    # wants the whole field on every rank (the reference writes one file, backproject.py:330).
    F_rows, d_rows, row0 = reduce_partials_sharded(F, d, F_store)
    out_rows = eng.finalize(F_rows, d_rows) if eng is not None else finalize_reference(F_rows, d_rows)
    out = gather_rows(out_rows, n) if gather else out_rows
    if verbose and rank == 0:
        print("Time taken for feature backprojection", time.time() - t0)  # backproject.py:171
    if return_partials:
        # F_rows/d: the SUMMED accumulators (this rank's row block of F starting at row0, and all of d)
        return out, F_rows, d, dict(stats, row0=row0)
    return out


def prune_mask(d: torch.Tensor) -> torch.Tensor:
    """utils.prune_by_gradients (utils.py:222-271) keeps Gaussians whose accumulated colour-gradient norm is
    > 0 over all views; that norm is 2/(3HW) * sum_v sum_p w, i.e. the mask is exactly d > 0."""
    return d > 0
