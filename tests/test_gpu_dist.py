"""-m gpu: bench.py's multi-rank control flow on the one GPU of the test box.  (The native map-sharded factor at world > 1 is
covered by the in-process transport in tests/test_gpu_shard_native.py / test_gpu_shard_fullsize.py, the protocol at world 2 over
gloo on CPU in test_dist_cpu.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_rank_control_flow_dry_run():
    """bench.py under torch.distributed.run with TWO ranks — the launch line the driver uses for the scaling runs — on the one
    GPU of the test box: MH_BENCH_DRYRUN=1 puts both ranks on device 0 over gloo (RCCL refuses two ranks on one device).  The
    numbers mean nothing; what is checked is that every multi-rank leg runs to the end: replica timing with its max-over-ranks
    reduction, one replay per rank, and ONE JSON line.  The map-sharded leg is the native one (RCCL inside the library), which two
    processes cannot run on one device: it must say so and leave `value` to the replicas."""
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
    assert "error" not in sh and "skipped" in sh, sh
    assert d["value_replica"] == d["value"] and "replicas" in d["metric_form"]
