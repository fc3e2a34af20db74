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


def _run_ranks(worker, world, timeout=600):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and f"OK {r}" in out, err[-2000:]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_layer_on_hip_backend_multi_rank(world):
    """Two / three ranks on the one GPU of the box (gloo collectives, HIP compute): sharded == unsharded oracle.  With three
    ranks a pose step can leave one rank without anything to send or receive while the other two trade points: the record
    exchange must still be entered (or skipped) by all of them together."""
    _run_ranks("dist_gpu_worker2.py", world)


def test_sharded_configs2_size_world2():
    """BASELINE configs[2] at full size on the one GPU of the box: 131 072-pt scan vs the ~50 M-pt map sharded over two ranks
    == the unsharded HIP factor on the full map."""
    _run_ranks("dist_gpu_worker3.py", 2, timeout=900)


def test_bench_two_rank_control_flow_dry_run():
    """bench.py under torch.distributed.run with TWO ranks — the launch line the driver uses for the scaling runs — on the one
    GPU of the test box: MH_BENCH_DRYRUN=1 puts both ranks on device 0 over gloo (RCCL refuses two ranks on one device).  The
    numbers mean nothing; what is checked is that every multi-rank leg runs to the end: replica timing with its max-over-ranks
    reduction, one replay per rank, the map-sharded leg (whose status histogram must be the unsharded one) and ONE JSON line."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MH_BENCH_DRYRUN="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["sequence_replay"]["n_ranks"] == 2 and min(d["sequence_replay"]["scans_per_s_per_rank"]) > 0
    sh = d["sharded"]
    assert "error" not in sh, sh
    assert sh["n_ranks"] == 2 and sh["scan_points_total"] == 131072 and sh["scan_points_max_per_rank"] < 131072
    assert sum(sh["status_hist"]) == 131072 and sh["status_hist"][8] > 80000
