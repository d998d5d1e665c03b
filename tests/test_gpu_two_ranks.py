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
