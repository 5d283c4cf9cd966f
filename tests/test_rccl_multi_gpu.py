"""The RCCL transport with MORE THAN ONE rank (comm.hip: grouped ncclSend / ncclRecv between x-neighbours, all-reduced
convergence sums).  Self-arming: runs whenever the box shows at least two GPUs — one rank per GPU under
torch.distributed.run, up to 4 — and skips on the single-GPU boxes; the loopback tests (test_dist_gpu.py) cover the same
World code on one GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _ngpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_ngpus() < 2, reason="needs at least two GPUs on the node (multi-rank RCCL)")
def test_slabs_over_rccl_match_the_undivided_domain():
    import socket

    n = min(_ngpus(), 4)
    with socket.socket() as sk:  # a free rendezvous port (the driver may have its own bench running next to the tests)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "rccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert all(f"RCCL_OK {k}" in r.stdout for k in range(n)), r.stdout[-2000:]
