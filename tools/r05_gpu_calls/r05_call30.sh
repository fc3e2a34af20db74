#!/bin/bash
# round-5 GPU call 30: photometric factor kernel with DPP wave sums instead of 43 (unary) / 118 (binary) fp64 shuffle butterflies
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c30
mkdir -p $O
export TMPDIR=/tmp
MH_FUZZ_EXTRA=100 timeout 1200 python -m pytest tests/test_gpu_photo.py tests/test_gpu_photo_fuzz.py tests/test_replay.py -q -m gpu -x > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
for p in 8 5; do
(cd /tmp && rm -rf /tmp/ptrace && MH_PATCH=$p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptrace -- python $R/tools/photo_trace.py > $R/$O/photo_trace_$p.log 2>&1)
python3 tools/save_stats.py /tmp/ptrace $O/photo_trace_kernel_stats_$p.csv; grep "photo_linearize" $O/photo_trace_kernel_stats_$p.csv
done
timeout 900 python -m pytest tests -q -m gpu -x -k "photo or replay or frontend" > $O/pytest_rel.log 2>&1; echo "rc $?" >> $O/pytest_rel.log; tail -n 2 $O/pytest_rel.log
