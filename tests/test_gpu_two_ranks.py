"""The N > 1 path with the real HIP engines: two FRESH child processes (torch.distributed.run) share the one GPU of the
box over a gloo process group and run the product driver; rank 0 checks the exchanged result against a single-process
run (tests/two_rank_worker.py).  The pytest process itself never re-execs."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_single_process(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "two_rank_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "TWO_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_two_ranks_bookkeeping_on_one_device(dev):
    """bench.py's N > 1 path (view sharding per rank, reduce-scatter of F + all-reduce of d inside the timed region, MAX /
    SUM over ranks, the post-run check on the reduced rows) with two fresh ranks that share the one GPU over gloo
    (--one-device: RCCL refuses two ranks on one device; the driver's real multi-GPU runs use nccl)."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--config", "C1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--dist-backend", "gloo", "--one-device"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["checked"]["ok"] is True
    assert j["config"]["overflow"] == 0 and j["value"] > 0
