"""Host-side logic that needs no GPU: synthetic generators, view sharding, SH basis, reduction semantics."""
import math
import os

import numpy as np
import pytest
import torch

import gsbp_amd
from gsbp_amd import synthetic as syn


def test_configs_mirror_baseline_json():
    c = syn.CONFIGS
    assert (c["C1"].n_gaussians, c["C1"].n_views, c["C1"].width, c["C1"].height, c["C1"].feat_dim) == (10_000, 4, 400, 300, 32)
    assert (c["C2"].n_gaussians, c["C2"].n_views, c["C2"].width, c["C2"].height, c["C2"].feat_dim) == (1_000_000, 200, 1600, 1060, 512)
    assert (c["C4"].n_gaussians, c["C4"].n_views, c["C4"].feat_dim) == (5_000_000, 300, 768)
    assert c["C5"].encoder_dim == 16 and c["C5"].feat_dim == 512


def test_scene_and_cameras_are_seeded_and_well_formed():
    cfg = syn.CONFIGS["T1"]
    a, b = syn.make_scene(cfg), syn.make_scene(cfg)
    assert all(torch.equal(a[k], b[k]) for k in a)
    means, quats, scales, opac = syn.activate(a)
    assert quats.shape == (cfg.n_gaussians, 4) and float(scales.min()) > 0 and 0 < float(opac.min()) < float(opac.max()) < 1
    assert abs(float(quats.norm(dim=1).mean()) - 1.0) > 0.1  # unnormalised on purpose (backproject.py:57)
    K = syn.intrinsics(cfg)
    assert int(K[0, 2] * 2) == cfg.width and int(K[1, 2] * 2) == cfg.height  # backproject.py:85-86
    vms = syn.make_cameras(cfg, n_views=7)
    R = vms[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(7, 3, 3), atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(7), atol=1e-5)
    c = -(R.transpose(1, 2) @ vms[:, :3, 3:])[:, :, 0]  # camera centres
    assert torch.all((c.norm(dim=1) > 3.1) & (c.norm(dim=1) < 3.9))
    origin_cam = vms[:, :3, 3]  # world origin in camera coordinates: straight ahead (+z)
    assert torch.all(origin_cam[:, 2] > 3.0) and float(origin_cam[:, :2].abs().max()) < 1e-4
    f = syn.make_feature_map(cfg, 3)
    assert f.shape == (cfg.height, cfg.width, cfg.feat_dim)
    assert torch.allclose(f.norm(dim=-1), torch.ones(cfg.height, cfg.width), atol=1e-5)


def test_view_shard_partitions_all_views():
    for world in (1, 2, 3, 8):
        got = sorted(v for r in range(world) for v in syn.view_shard(200, r, world))
        assert got == list(range(200))
        sizes = [len(syn.view_shard(200, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_sh_basis_degree0_and_symmetry():
    from ref_sh import spherical_harmonics
    n = 32
    g = torch.Generator().manual_seed(0)
    dirs = torch.randn(n, 3, generator=g)
    coeffs = torch.randn(n, 16, 3, generator=g)
    c0 = spherical_harmonics(0, dirs, coeffs[:, :1])
    assert torch.allclose(c0, 0.28209479177387814 * coeffs[:, 0])
    # scaling the direction does not change the colour; odd bands flip sign under d -> -d
    c3 = spherical_harmonics(3, dirs, coeffs)
    assert torch.allclose(c3, spherical_harmonics(3, 5.0 * dirs, coeffs), atol=1e-5)
    only1 = torch.zeros_like(coeffs)
    only1[:, 1:4] = coeffs[:, 1:4]
    assert torch.allclose(spherical_harmonics(3, dirs, only1), -spherical_harmonics(3, -dirs, only1),
                          atol=1e-6)


def test_prune_mask_is_positive_denominator():
    d = torch.tensor([0.0, 1e-9, 3.0])
    assert gsbp_amd.prune_mask(d).tolist() == [False, True, True]  # utils.py:257 keeps grads > 0


def test_create_feature_field_cpu_injection_matches_direct(orc):
    """The driver loop (sharding, reduction modes, encoder) exercised on CPU through the view_fn injection point."""
    cfg = syn.CONFIGS["T0"]
    means, quats, scales, opac = syn.activate(syn.make_scene(cfg))
    K, vms = syn.intrinsics(cfg), syn.make_cameras(cfg)
    feats = [syn.make_feature_map(cfg, v) for v in range(cfg.n_views)]
    h = [t.numpy() for t in (means, quats, scales, opac)]
    for reduction in ("sum", "mean"):
        sf = 1.0 if reduction == "sum" else 1.0 / (cfg.height * cfg.width * cfg.feat_dim)
        sd = 1.0 if reduction == "sum" else 1.0 / (cfg.height * cfg.width * 3)
        acc = {}

        def view_fn(v, f, acc=acc):
            acc.setdefault("F", np.zeros((cfg.n_gaussians, cfg.feat_dim), np.float64))
            acc.setdefault("d", np.zeros(cfg.n_gaussians, np.float64))
            orc.backproject_view(*h, vms[v].numpy(), K.numpy(), cfg.width, cfg.height, f.numpy(), acc["F"], acc["d"])

        out = gsbp_amd.create_feature_field(means, quats, scales, opac, vms, K, cfg.width, cfg.height,
                                            lambda v: feats[v], cfg.feat_dim, reduction=reduction, view_fn=view_fn)
        assert float(out.abs().max()) == 0.0  # injected view_fn accumulates elsewhere: driver's own F stays zero
        ref, _, _, _ = orc.backproject_oracle(*h, vms.numpy(), K.numpy(), cfg.width, cfg.height,
                                              lambda v: feats[v].numpy(), cfg.feat_dim, reduction=reduction)
        mine = orc.finalize(np.ascontiguousarray(acc["F"] * sf), np.ascontiguousarray(acc["d"] * sd))
        assert np.abs(mine - ref).max() < 1e-6


def test_nearest_index_matches_torch_interpolate():
    """The index maps gwbp_scatter_upsampled consumes must be F.interpolate(mode="nearest")'s (backproject.py:244-248)."""
    for n_in, n_out in ((64, 1060), (64, 1600), (16, 224), (37, 300), (5, 5), (7, 3), (1, 9), (64, 1297), (3, 1000)):
        src = torch.arange(n_in, dtype=torch.float32).reshape(1, 1, 1, n_in)
        ref = torch.nn.functional.interpolate(src, size=(1, n_out), mode="nearest")[0, 0, 0].to(torch.int32)
        assert torch.equal(gsbp_amd.nearest_index(n_in, n_out), ref), (n_in, n_out)


def test_bilinear_index_matches_torch_interpolate():
    """The maps gwbp_scatter_bilinear consumes, applied on the CPU, must reproduce F.interpolate(mode="bilinear",
    align_corners=False) (backproject.py:110-112)."""
    g = torch.Generator().manual_seed(3)
    for (h, w), (H, W) in (((24, 24), (106, 160)), ((7, 5), (33, 64)), ((16, 9), (16, 9)), ((3, 4), (10, 2)), ((1, 6), (5, 19))):
        low = torch.randn(h, w, 6, generator=g)
        ref = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                              align_corners=False)[0].permute(1, 2, 0)
        (y0, ly), (x0, lx) = gsbp_amd.bilinear_index(h, H), gsbp_amd.bilinear_index(w, W)
        y0, x0 = y0.long(), x0.long()
        y1, x1 = (y0 + 1).clamp(max=h - 1), (x0 + 1).clamp(max=w - 1)
        h1, w1 = ly[:, None, None], lx[None, :, None]
        h0, w0 = 1 - h1, 1 - w1
        a, b = low[y0][:, x0], low[y0][:, x1]
        c, d = low[y1][:, x0], low[y1][:, x1]
        mine = h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
        assert float((mine - ref).abs().max()) <= 2e-5, ((h, w), (H, W))  # ATen vectorises the index arithmetic differently: ~5e-6


def test_bench_starts_its_own_ranks_for_gpus_n(monkeypatch):
    """`python bench.py --gpus N` as a plain process: bench.self_launch() must start N fresh ranks under
    torch.distributed.run on 127.0.0.1 with the SAME arguments, before importing torch in the parent, and hand back the
    launcher's exit code (the GPU-side end-to-end run is tests/test_gpu_two_ranks.py)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)  # defines main / self_launch, runs nothing
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--total-views", "9"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    # fewer GPUs than ranks: a one-line refusal, nothing is launched (round 6: no hang in init_process_group)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "--gpus 4" in str(e.value.code) and "2 GPU(s)" in str(e.value.code) and not seen
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--total-views", "9"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_finds_the_committed_counters_of_the_kernel_it_timed(tmp_path):
    """bench.py's `roofline.traffic` comes from profiles/traffic.json and only when that file's passes profiled the kernel
    the run timed: the file holds rocprof's names (namespace, template arguments), the bench its own short ones."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"C2": {"scatter_kernel": "gwbp::k_scatter_wide<false>", "scatter_hbm_bytes_per_launch": 12.0,
                                    "scatter_valu_wave_instructions": 5.0, "source": "x"},
                             "C5": {"scatter_kernel": "gwbp::k_blend<2>", "scatter_hbm_bytes_per_launch": 3.0}}))
    assert bench.committed_traffic("C2", "k_scatter_wide", str(f)) == (12.0, "profiles/traffic.json (x)", 5.0)
    assert bench.committed_traffic("C2", "k_scatter_full", str(f)) == (None, None, None)   # another kernel's counters
    assert bench.committed_traffic("C5", "k_blend<kFused> (blend + scatter in one kernel, no weight store)", str(f))[0] == 3.0
    assert bench.committed_traffic("C4", "k_scatter_full", str(f)) == (None, None, None)   # no passes for this config
    assert bench.committed_traffic("C2", "k_scatter_wide", str(tmp_path / "absent.json")) == (None, None, None)
    # the committed file matches the kernels the default pipeline picks at the three bench configurations
    for cfg, kern in (("C2", "k_scatter_wide"), ("C4", "k_scatter_full"), ("C5", "k_blend<kFused>")):
        assert bench.committed_traffic(cfg, kern)[0] > 0, cfg


def test_one_camera_batch_indexes_without_a_select_backward():
    """rasterization() hands a one-camera render back as [1,H,W,D]; the reference back-propagates through
    `output_for_grad[0]` (backproject.py:127-129).  That index must return the [H,W,D] tensor the batch is a view of --
    autograd's select backward would zero-fill and copy a whole [1,H,W,D] per view -- with the gradients of every way of
    indexing unchanged, and everything else a plain tensor."""
    from gsbp_amd.rasterization import one_camera_batch
    w = torch.randn(5, 4, 3)

    def fresh():
        leaf = torch.randn(5, 4, 3, requires_grad=True)
        render = leaf * 2
        return leaf, render, one_camera_batch(render)

    leaf, render, batch = fresh()
    assert batch.shape == (1, 5, 4, 3) and batch.requires_grad and batch[0] is render and batch[-1] is render
    assert batch.data_ptr() == render.data_ptr()
    for index in (lambda b: b[0], lambda b: b[-1], lambda b: b[0:1], lambda b: b, lambda b: b.squeeze(0), lambda b: b[0, :, :],
                  lambda b: b.sum(0), lambda b: torch.cat([b, b])[1]):
        leaf, _, batch = fresh()
        (index(batch) * w).sum().backward()
        assert torch.equal(leaf.grad, 2 * w)
    leaf, render, batch = fresh()
    (batch[0] * w).sum().backward()  # the reference's form: no SelectBackward between the product and the render
    assert "Select" not in type(batch[0].grad_fn).__name__ and batch[0].grad_fn is render.grad_fn
    for plain in (batch * 1, batch[0:1], batch.detach(), batch.clone(), batch.sum(), batch[0, 1]):
        assert type(plain) is torch.Tensor
    with pytest.raises(IndexError):
        batch[1]
    with torch.no_grad():  # one storage: an in-place change of the batch is a change of the camera's render
        batch.mul_(0)
    assert float(render.detach().abs().max()) == 0.0


def test_zero_render_is_a_fresh_tensor_per_call_and_keeps_autograd_edges_apart():
    """the cached all-zero render must not be ONE tensor object handed out by every harvest forward --
    autograd hangs the producing node on the returned object, so a second apply() would re-point the first call's output at
    the second call's node.  The helper returns a new object over the same storage each time; a toy Function that returns it
    twice before either backward gets both gradients."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]  # (the package attribute of that name is the function)
    rz.invalidate_zero_table_cache()
    a = rz._zero_render("cpu", 2, 3, 4)
    b = rz._zero_render("cpu", 2, 3, 4)
    assert a is not b and a.data_ptr() == b.data_ptr() and a.shape == (2, 3, 4)
    a.add_(1.0)  # an in-place write through a handed-out alias is noticed: the next render is zero again
    c = rz._zero_render("cpu", 2, 3, 4)
    assert float(c.abs().max()) == 0.0

    class Harvest(torch.autograd.Function):
        @staticmethod
        def forward(ctx, table, k):
            ctx.k = k
            return rz._zero_render("cpu", 2, 3, 4)

        @staticmethod
        def backward(ctx, g):
            return torch.full((5,), float(ctx.k)) * g.sum(), None

    t = torch.zeros(5, requires_grad=True)
    o1, o2 = Harvest.apply(t, 1.0), Harvest.apply(t, 20.0)
    assert o1 is not o2 and o1.grad_fn is not o2.grad_fn
    (o1.sum() + o2.sum()).backward()
    assert torch.equal(t.grad, torch.full((5,), 24.0 * 21.0))


def test_direct_grad_path_is_taken_only_where_it_cannot_be_observed():
    """the drop-in's backward adds straight into leaf.grad only under a plain .backward() on a leaf without
    hooks; torch.autograd.grad() and hooked leaves must get a returned gradient."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]  # (the package attribute of that name is the function)
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.leaf, ctx.node = x, rz._accumulate_node(x)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(rz._grad_is_unobserved(ctx.leaf, ctx.node))
            return g * 2

    x = torch.ones(3, requires_grad=True)
    Probe.apply(x).sum().backward()
    torch.autograd.grad(Probe.apply(x).sum(), x)
    Probe.apply(x).sum().backward(inputs=[x])
    h = x.register_hook(lambda g: g)
    Probe.apply(x).sum().backward()
    h.remove()
    h = x.register_post_accumulate_grad_hook(lambda t: None)
    Probe.apply(x).sum().backward()
    h.remove()
    Probe.apply(x).sum().backward()
    assert seen == [True, False, True, False, False, True]


def test_zero_table_verdict_is_rechecked_on_a_sample_and_can_be_invalidated():
    """writes through .data do not move the version counter; a dense one is caught by the per-call sample,
    a sparse one by invalidate_zero_table_cache()."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]  # (the package attribute of that name is the function)
    rz.invalidate_zero_table_cache()
    t = torch.zeros(1000, 8, requires_grad=True)
    assert rz._is_zero_table(t) and rz._is_zero_table(t)
    v = t._version
    t.data.add_(1.0)
    assert t._version == v          # the blind spot of a (tensor, version) key ...
    assert not rz._is_zero_table(t)  # ... which the sample closes for dense writes
    t.data.zero_()
    assert not rz._is_zero_table(t)  # (cached "non-zero" at this version: the safe side -- a full render, never a wrong one)
    rz.invalidate_zero_table_cache()
    assert rz._is_zero_table(t)
    t.data[1, 0] = 3.0               # one row outside the sample: needs the explicit invalidation
    rz.invalidate_zero_table_cache()
    assert not rz._is_zero_table(t)


def test_view_pipeline_refuses_to_run_on_shared_hardware_queues(monkeypatch):
    """GPU_MAX_HW_QUEUES was set by bench.py only, so the product CLI and library callers ran the pipeline on the
    runtime's default 4 queues.  The package now asks for 8 when it is imported; if that came too late (runtime already up) or
    the caller asked for fewer, ViewPipeline says so instead of silently running front + scatter back to back."""
    from gsbp_amd import _lib
    assert os.environ.get("GPU_MAX_HW_QUEUES") is not None and _lib.hw_queues_ok()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    monkeypatch.setattr(_lib, "_QUEUES_LATE", True)  # the runtime was up before the package could ask: refuse
    assert not _lib.hw_queues_ok() and _lib.hw_queues_late()
    with pytest.raises(gsbp_amd.GwbpError, match="GPU_MAX_HW_QUEUES"):
        gsbp_amd.ViewPipeline(100, 64, 48, "cpu")
    monkeypatch.setattr(_lib, "_QUEUES_LATE", False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")     # somebody chose fewer: say so, do not refuse
    assert not _lib.hw_queues_ok() and not _lib.hw_queues_late()
    with pytest.warns(RuntimeWarning, match="GPU_MAX_HW_QUEUES"):
        try:
            gsbp_amd.ViewPipeline(100, 64, 48, "cpu")
        except Exception:  # (no HIP device here: the engines behind the guard cannot be created)
            pass


def test_private_torch_apis_of_the_harvest_shortcut_exist():
    """the drop-in's harvest shortcut pattern-matches through private torch entry points.  A torch release that
    renames one must fail HERE, by name, not as a silent slow-down (the package switches the shortcut off at import then)."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]
    assert rz.missing_private_apis() == [], f"torch {torch.__version__} lacks {rz.missing_private_apis()}"
    assert rz._HARVEST_SHORTCUT


def test_zero_table_cache_sees_a_swapped_storage():
    """`t.data = other` keeps the tensor object AND its version counter: only the storage address tells."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]
    rz.invalidate_zero_table_cache()
    t = torch.zeros(1000, 8, requires_grad=True)
    assert rz._is_zero_table(t)
    other = torch.zeros(1000, 8)
    other[3, 1] = 2.0  # a row the strided sample does not look at
    v = t._version
    t.data = other
    assert t._version == v and not rz._is_zero_table(t)
    rz.invalidate_zero_table_cache()


def test_harvest_sum_backward_is_differentiable_under_create_graph():
    """`(render * feats).sum().backward(create_graph=True)` used to raise through once_differentiable; the shortcut's backward
    now returns g * feats WITH history there.  CPU check on the Function itself."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]
    render = torch.zeros(3, 4, 2, requires_grad=True)
    feats = torch.randn(3, 4, 2)
    y = rz._HarvestSum.apply(render, feats, 1.0)
    (g,) = torch.autograd.grad(y, render, create_graph=True)
    assert torch.equal(g, feats) and g.requires_grad is False or torch.allclose(g, feats)
    (g2,) = torch.autograd.grad(rz._HarvestSum.apply(render, feats, 0.5), render)
    assert torch.allclose(g2, 0.5 * feats)


def test_harvest_render_shortcut_and_its_fallbacks():
    """The render of an all-zero differentiable table recognises the reference's harvest statement
    `(render * feats).sum().backward()` (backproject.py:127-129) and hands feats to the rasteriser's backward without computing
    the product, its sum or the product's gradient; every other use falls back to the literal computation.  CPU check of the
    tensor subclasses against a toy render that is attached to a graph."""
    import sys
    rz = sys.modules["gsbp_amd.rasterization"]
    leaf = torch.zeros(4, 5, 3, requires_grad=True)
    f = torch.randn(4, 5, 3)

    def render():
        return (leaf * 1.0).as_subclass(rz._HarvestRender)

    def grad_of(loss):
        leaf.grad = None
        loss.backward()
        return leaf.grad.clone()

    p = render() * f
    assert type(p) is rz._HarvestProduct and p.shape == f.shape and p.requires_grad and p.numel() == f.numel()  # (metadata only: no kernel)
    s = p.sum()
    assert type(s) is torch.Tensor and "HarvestSum" in type(s.grad_fn).__name__ and float(s.detach()) == 0.0
    assert torch.equal(grad_of(s), f)
    assert torch.equal(grad_of((f * render()).sum()), f)                       # __rmul__
    assert torch.allclose(grad_of(render().mul(f).sum() * 2.5), 2.5 * f)       # an incoming scalar other than one
    m = (render() * f).mean()                                                  # the dino variant's .mean(), backproject.py:263
    assert "HarvestSum" in type(m.grad_fn).__name__ and torch.allclose(grad_of(m), f / f.numel())
    # fallbacks: the literal computation, same gradients
    assert torch.allclose(grad_of((render() * f).mean(dim=(0, 1)).sum()), f / (f.shape[0] * f.shape[1]))
    assert torch.allclose(grad_of((render() * f).sum(dim=0).sum()), f)
    g = grad_of((render() * f)[..., :2].sum())
    assert torch.equal(g[..., :2], f[..., :2]) and float(g[..., 2].abs().max()) == 0.0
    fg = f.clone().requires_grad_(True)
    assert torch.equal(grad_of((render() * fg).sum()), f) and float(fg.grad.abs().max()) == 0.0   # feats with grad: d/dfeats = render = 0
    assert grad_of((render() * f[:, :, :1]).sum()).shape == leaf.shape         # broadcasting
    assert type(render() + 1.0) is torch.Tensor and type(render()[None]) is torch.Tensor
    with torch.no_grad():
        assert type(render() * f) is torch.Tensor
    # two statements on one render accumulate like any two uses
    r = render()
    assert torch.allclose(grad_of((r * f).sum() + (r * (2 * f)).sum()), 3 * f)
    # the placeholder inside a LIST argument is the literal product too (never the detached zeros it physically is)
    st = torch.stack([render() * f, render() * (2 * f)])
    assert st.requires_grad and torch.allclose(grad_of(st.sum()), 3 * f)
    # feats written in place between the statement and its backward: an error, as for the literal product's saved operand
    f2 = f.clone()
    s = (render() * f2).sum()
    f2.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        s.backward()


def test_cli_accepts_every_parameter_of_the_reference_main():
    """The reference's main() (backproject.py:301-311) takes data_dir, checkpoint, results_dir, rasterizer, data_factor,
    feature_field_batch_count, run_feature_field_on_cpu, feature.  A caller that passes any of them must not get an argparse
    error from run_backproject.py (the two no-op parameters were missing; --help itself crashed on a bare %)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "run_backproject.py"), "--data-dir", "x", "--checkpoint", "y",
                        "--results-dir", "z", "--rasterizer", "gsplat", "--data-factor", "4", "--feature-field-batch-count", "3",
                        "--run-feature-field-on-cpu", "--feature", "dino", "--help"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    for flag in ("--feature-field-batch-count", "--run-feature-field-on-cpu", "--no-run-feature-field-on-cpu"):
        assert flag in r.stdout


def test_token_geometry_rule_follows_the_exact_index_maps():
    """Engine.token_geometry_ok decides (on the host, from PyTorch's own fp32 nearest rule) whether every 16 x 16 tile of the view
    sees at most 2 x 2 texels of a low-resolution map -- the precondition of gwbp_blend_tokens.  The dino script's 64 x 64 tokens at
    the garden scene's 1600 x 1060 qualify, the lseg script's 480 x 480 map does not; the rule is checked against a brute-force
    walk over the index maps, including sizes where 16 * n_in <= n_out fails by one texel."""
    import itertools
    from gsbp_amd import Engine, nearest_index
    assert Engine.token_geometry_ok(64, 64, 1060, 1600)
    assert not Engine.token_geometry_ok(480, 480, 1060, 1600)
    assert Engine.token_geometry_ok(1, 1, 9, 13) and Engine.token_geometry_ok(8, 12, 136, 200)
    assert not Engine.token_geometry_ok(13, 17, 136, 200)
    for n_in, n_out in itertools.product((1, 2, 3, 5, 8, 9, 12, 13, 33, 66, 67), (1, 15, 16, 17, 136, 200, 1060)):
        m = nearest_index(n_in, n_out).tolist()
        assert all(b >= a for a, b in zip(m, m[1:])) and m[-1] <= n_in - 1  # non-decreasing, inside the map
        brute = all(m[min(t + 15, n_out - 1)] - m[t] <= 1 for t in range(0, n_out, 16))
        assert Engine.token_geometry_ok(n_in, 1, n_out, 16) == brute, (n_in, n_out)
