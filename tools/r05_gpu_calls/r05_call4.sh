#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_overlap.py tests/test_gpu_batch.py tests/test_gpu_configs1.py tests/test_gpu_reported_basis.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python tools/k3_time.py > $O/k3_time_new.json 2> $O/k3_time_new.err
SYNC_PROBE_CALLS=100 timeout 600 python tools/sync_probe.py > $O/sync_probe_new.json 2> $O/sync_probe_new.err
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1
MH_TL_POINTS=24576 timeout 300 python tools/timeline.py > $O/timeline_24k.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
ls -la $O
