"""run_backproject.py, the counterpart of the reference's main() (backproject.py:301-336): prune_by_gradients ->
(test_proper_pruning when the checkpoint has SH colours) -> feature field of the PRUNED scene -> features_<kind>.pt.
Runs the CLI as a child process on the synthetic C1 scene, with and without --no-prune.  A pruned Gaussian has no weight
in any view, but it may still have been the Gaussian that TERMINATES pixels (T' <= 1e-4 stops the pixel without counting
it): without it those pixels run one Gaussian further, i.e. the kept Gaussians' sums move by weights of order 1e-4 --
the same happens in the reference, which also builds on the pruned scene.  So the rows agree closely, not exactly."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, *flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_backproject.py"), "--synthetic", "C1", "--results-dir",
                        str(tmp), *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_cli_prunes_first_like_the_reference_main(dev, tmp_path):
    a, b = tmp_path / "pruned", tmp_path / "all"
    out = _run(a)
    assert "Total splats 10000" in out and "Remaining" in out and "Time taken for feature backprojection" in out
    _run(b, "--no-prune")
    keep = torch.load(a / "prune_mask.pt")
    fa, fb = torch.load(a / "features_lseg.pt"), torch.load(b / "features_lseg.pt")
    assert keep.dtype == torch.bool and keep.shape == (10000,) and not (b / "prune_mask.pt").exists()
    assert 0 < int(keep.sum()) < 10000 and fa.shape == (int(keep.sum()), 32) and fb.shape == (10000, 32)
    err = (fa - fb[keep]).abs().max(dim=1).values  # unit rows
    assert float(err.median()) <= 1e-6 and float(err.quantile(0.99)) <= 2e-3 and float(err.max()) <= 0.1
    assert float(fb[~keep].abs().max()) == 0.0          # never-seen Gaussians: 0/0 -> NaN -> 0 (backproject.py:169)


def test_cli_prune_by_product_is_one_sweep_with_the_same_mask(dev, tmp_path):
    """--prune-by-product ("the mask comes free from the fused kernel"): the field is built once on ALL
    Gaussians, keep = d > 0 from the same denominators, the pruned rows are dropped.  Same mask as the separate sweep bit for bit;
    the kept rows are those of the --no-prune build exactly, and those of the prune-first build up to the terminating-Gaussian
    effect described at the top of this file."""
    a, b, c = tmp_path / "first", tmp_path / "byproduct", tmp_path / "all"
    _run(a)
    out = _run(b, "--prune-by-product")
    _run(c, "--no-prune")
    assert "Total splats 10000" in out and "Remaining" in out
    ka, kb = torch.load(a / "prune_mask.pt"), torch.load(b / "prune_mask.pt")
    assert torch.equal(ka, kb)
    fa, fb, fc = (torch.load(x / "features_lseg.pt") for x in (a, b, c))
    assert fb.shape == fa.shape == (int(ka.sum()), 32)
    assert float((fb - fc[kb]).abs().max()) <= 1e-5      # the same build, sliced (atomic order only)
    err = (fa - fb).abs().max(dim=1).values
    assert float(err.median()) <= 1e-6 and float(err.quantile(0.99)) <= 2e-3 and float(err.max()) <= 0.1


def test_cli_two_ranks_shard_sweep_check_and_build(dev, tmp_path):
    """The CLI under a process group (two fresh ranks sharing the one GPU over gloo): the prune sweep and the field build are
    sharded by view, the weight sums all-reduced -- same mask, same field as the single-process run."""
    import socket
    a, b = tmp_path / "one", tmp_path / "two"
    _run(a)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "run_backproject.py"), "--synthetic", "C1",
                        "--results-dir", str(b), "--dist-backend", "gloo", "--one-device"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("Total splats 10000") == 1  # rank 0 reports
    ka, kb = torch.load(a / "prune_mask.pt"), torch.load(b / "prune_mask.pt")
    assert torch.equal(ka, kb)
    fa, fb = torch.load(a / "features_lseg.pt"), torch.load(b / "features_lseg.pt")
    assert fa.shape == fb.shape and float((fa - fb).abs().max()) <= 1e-4


def test_cli_dino_shaped_map_goes_through_token_space(dev, tmp_path):
    """`run_backproject.py --synthetic T1D` (an 8 x 12 x 256 token map, nearest, .mean(): the dino script's shape at test size):
    the CLI hands the network-resolution map to the driver with upsample="nearest", which takes the token-space kernels; the saved
    field equals the pixel-slab path's (token_space=False) run in-process on the same seeded inputs."""
    import torch
    import gsbp_amd
    from gsbp_amd import synthetic as syn
    out_dir = tmp_path / "tok"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_backproject.py"), "--synthetic", "T1D", "--results-dir", str(out_dir),
                        "--feature", "dino", "--no-prune"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = torch.load(out_dir / "features_dino.pt")
    cfg = syn.CONFIGS["T1D"]
    g = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
    ref = gsbp_amd.create_feature_field(*g, syn.make_cameras(cfg).to(dev), syn.intrinsics(cfg).to(dev), cfg.width, cfg.height,
                                        lambda v: syn.make_feature_map(cfg, v, device=dev), cfg.feat_dim, reduction="mean",
                                        upsample="nearest", token_space=False)
    assert tuple(got.shape) == (cfg.n_gaussians, cfg.feat_dim)
    assert float((got.to(dev) - ref).abs().max()) <= 2e-5
