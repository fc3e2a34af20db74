#!/bin/bash
# round-5 GPU call 1: fused sync path tests, sync latency probe, K3 variants, timelines, API trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c1
O=gpurun_out/c1
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_overlap.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python tools/sync_probe.py > $O/sync_probe.json 2> $O/sync_probe.err
timeout 300 python tools/k3_time.py > $O/k3_time_base.json 2> $O/k3_time_base.err
MH_LIB_OVERRIDE=$PWD/mimosa_amd/lib/variants/interleave.so timeout 300 python tools/k3_time.py > $O/k3_time_interleave.json 2> $O/k3_time_interleave.err
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1
MH_TL_POINTS=24576 timeout 300 python tools/timeline.py > $O/timeline_24k.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SYNC_PROBE_CALLS=40 timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/trace1 -- python $GRAFT_REPO_ROOT/tools/sync_probe.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
for f in $(ls /tmp/trace1/*/*_hip_api_trace.csv /tmp/trace1/*/*_kernel_trace.csv 2>/dev/null); do cp $f $GRAFT_REPO_ROOT/$O/; done
ls -la $GRAFT_REPO_ROOT/$O
tail -3 $GRAFT_REPO_ROOT/$O/pytest.log
