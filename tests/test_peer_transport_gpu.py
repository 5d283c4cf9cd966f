"""The xGMI peer-direct transport (comm_peer.hip) between real processes: flagged stores into hipIpc-mapped windows instead of
RCCL calls.  Unlike RCCL it accepts two ranks on one GPU, so this is the multi-PROCESS decomposed run the single-GPU boxes can
execute: two ranks (three where the box has the GPUs to spare or not — the ranks share devices round-robin), the transport's
own self-test with slots shorter than the messages, then 8 steps against the undivided domain."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("nranks", [2, 3])
def test_slabs_over_the_peer_transport_match_the_undivided_domain(nranks):
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "peer_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert all(f"PEER_OK {k}" in r.stdout for k in range(nranks)), r.stdout[-2000:]
