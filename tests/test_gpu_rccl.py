"""RCCL executes on the device: the sharded job's pack / gather / scatter (cpupathtrace_amd/sharding.py, the N > 1 path of bench.py) with
backend `nccl` in a world of one rank on the leased GPU.  The N = 8 run itself is the driver's; this pins that the collective path works with
device tensors end to end (librccl loaded, chunks stay in HBM, frame bit-identical with processJob)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_gather_over_rccl_on_the_device():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    # a fresh interpreter: nothing that has initialised the GPU is re-executed, the child starts from scratch
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py"), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["rccl_mapped"], "librccl is not among the libraries the process mapped"
    assert out["recv_on_device"], "the gathered chunks must stay in device memory"
    assert out["identical"], "the frame that came back through the gather differs from processJob's"
    assert out["checksum_ok"] and out["nonzero_pixels"] > 0 and out["gather_bytes"] == 96 * 80 * 16
