"""-m gpu: the sharding layer (mimosa_amd/dist.py) driving the HIP backend through RCCL ("nccl") on one
GPU (world size 1: the only size available to the test box; world 2 runs on CPU/gloo in
test_dist_cpu.py).  Runs in a fresh process so that torch initialises the device first, as in bench.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_layer_on_hip_backend_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
