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


def test_sharded_layer_on_hip_backend_world2():
    """Two ranks on the one GPU of the box (gloo collectives, HIP compute): sharded == unsharded oracle."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker2.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and f"OK {r}" in out, err[-2000:]


def test_sharded_configs2_size_world2():
    """BASELINE configs[2] at full size on the one GPU of the box: 131 072-pt scan vs the ~50 M-pt map sharded over two ranks
    == the unsharded HIP factor on the full map."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker3.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and f"OK {r}" in out, err[-2000:]
