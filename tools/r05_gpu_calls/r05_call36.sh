#!/bin/bash
# round-5 GPU call 36: the projection's front in stage A's launch, its back in stage B's
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c36
mkdir -p $O
export TMPDIR=/tmp
MH_FUZZ_EXTRA=100 timeout 1200 python -m pytest tests/test_gpu_photo.py tests/test_gpu_photo_fuzz.py tests/test_replay.py tests/test_scan_frontend.py -q -m gpu -x > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
MH_PHOTO_UNFUSED=1 timeout 600 python -m pytest tests/test_gpu_photo.py -q -m gpu -x > $O/pytest_photo_unfused.log 2>&1; echo "rc $?" >> $O/pytest_photo_unfused.log; tail -n 2 $O/pytest_photo_unfused.log
for k in 1 2 3; do timeout 300 python tools/photo_resident_time.py 2> $O/photo_resident.err | tee -a $O/photo_resident.jsonl; done
(cd /tmp && rm -rf /tmp/prof_photo && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_photo -- python $R/tools/photo_resident_time.py > /dev/null 2>&1)
python3 tools/save_stats.py /tmp/prof_photo $O/photo_kernel_stats.csv; grep "photo_" $O/photo_kernel_stats.csv | cut -c1-100
