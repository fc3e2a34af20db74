#!/bin/bash
# round-5 GPU call 15: K4 with a decomposition wave — parity (incl. synchronous == pipelined to the bit), timeline, cold kernel stats, sync probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c15
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_overlap.py tests/test_gpu_batch.py tests/test_gpu_configs1.py tests/test_gpu_reported_basis.py tests/test_gpu_fuzz.py tests/test_gpu_host_cpp.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1; grep "^K4" $O/timeline_131k.txt
cd /tmp && export TMPDIR=/tmp
COLD_PROBE_CALLS=120 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/coldstats -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe.log 2>&1
python3 $R/tools/save_stats.py /tmp/coldstats $R/$O/cold_kernel_stats.csv; grep icp_ $R/$O/cold_kernel_stats.csv | cut -c1-120
cd $R
SYNC_PROBE_CALLS=200 timeout 600 python tools/sync_probe.py > $O/sync_probe.json 2> $O/sync_probe.err; cut -c1-420 $O/sync_probe.json
