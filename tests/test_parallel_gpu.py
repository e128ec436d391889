"""The multi-rank training step on real device code: two ranks share GPU 0 and sum their gradients through gloo
(N2M_DIST_BACKEND=gloo; on the 8-GPU node the same code runs one rank per GPU over RCCL).  Covers the FusedAdamAMP multi-rank path:
per-table collectives in their own dtype, the persistent dW buffer + found_inf flag in the flat bucket, 1/world folded into the
seed gradient, lock-step GradScaler decisions."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_stay_bit_identical_and_learn():
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tools", "dist_check.py"), "40"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
