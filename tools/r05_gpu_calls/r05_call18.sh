#!/bin/bash
# round-5 GPU call 18: four-launch photometric preprocess (frame stamps, 1024-thread stage workgroups) + communicator registry
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c18
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_photo.py tests/test_gpu_shard_native.py -q -m gpu -x > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
MH_PHOTO_UNFUSED=1 timeout 600 python -m pytest tests/test_gpu_photo.py -q -m gpu -x > $O/pytest_photo_unfused.log 2>&1; echo "rc $?" >> $O/pytest_photo_unfused.log; tail -n 2 $O/pytest_photo_unfused.log
timeout 300 python tools/photo_resident_time.py > $O/photo_resident.json 2> $O/photo_resident.err; cat $O/photo_resident.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_photo -- python $OLDPWD/tools/photo_resident_time.py > /dev/null 2>&1)
python tools/save_stats.py $O/prof_photo $O/photo_kernel_stats.csv
head -12 $O/photo_kernel_stats.csv | cut -c1-160
rm -rf $O/prof_photo
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "rc $?" >> $O/pytest_full.log; tail -n 3 $O/pytest_full.log
