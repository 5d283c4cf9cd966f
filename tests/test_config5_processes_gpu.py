"""BASELINE config 5 (8 x 10^6 particles as eight x-slabs) between eight REAL processes over the xGMI peer-direct transport, the
processes sharing the box's GPU(s): the multi-process counterpart of tests/test_config5_gpu.py (eight host threads over the
loopback).  NOT YET RUN ON HARDWARE (written when the round's GPU budget was spent): first thing to run next round, with
SALVA_CONFIG5_SIDE=40 before the full size."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_config5_eight_processes_over_the_peer_transport():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "config5_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "CONFIG5_OK" in r.stdout, r.stdout[-2000:]
