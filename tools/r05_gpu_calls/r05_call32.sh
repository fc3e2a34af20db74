#!/bin/bash
# round-5 GPU call 32: photometric factor kernel with the LDS altitude table and the pose-index guess: tests, kernel time
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c32
mkdir -p $O
export TMPDIR=/tmp
MH_FUZZ_EXTRA=100 timeout 1200 python -m pytest tests/test_gpu_photo.py tests/test_gpu_photo_fuzz.py tests/test_replay.py -q -m gpu -x > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
for p in 8 5; do
(cd /tmp && rm -rf /tmp/ptrace && MH_PATCH=$p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptrace -- python $R/tools/photo_trace.py > $R/$O/photo_trace_$p.log 2>&1)
python3 tools/save_stats.py /tmp/ptrace $O/photo_trace_kernel_stats_$p.csv; grep "photo_linearize" $O/photo_trace_kernel_stats_$p.csv
done
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "rc $?" >> $O/pytest_full.log; tail -n 3 $O/pytest_full.log
