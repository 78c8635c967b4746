"""Data-parallel training and sharded sampling with two real ranks that share the one GPU (gloo carries the
collectives through the host; every kernel is the HIP path): see tests/dp_worker.py for what is checked."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_data_parallel_matches_the_single_process_reference():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DP_WORKER_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-6000:])
