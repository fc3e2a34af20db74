#!/bin/bash
# round-5 GPU call 35: closing soak of the final build under the allocation checker
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c35
mkdir -p $O
export MH_ALLOC_CHECK=1
timeout 2400 python -m pytest tests -q -m gpu > $O/full.log 2>&1; echo "rc $?" >> $O/full.log; tail -n 2 $O/full.log
for i in 1 2 3; do
  MH_FUZZ_EXTRA=300 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_shard_native.py tests/test_gpu_overlap.py tests/test_gpu_photo.py tests/test_gpu_photo_fuzz.py tests/test_zz_alloc_check.py -q -m gpu > $O/fuzz_$i.log 2>&1; echo "rc $?" >> $O/fuzz_$i.log; tail -n 2 $O/fuzz_$i.log
done
timeout 1200 python tools/replay_soak.py 12 > $O/replay_soak.log 2>&1; tail -n 1 $O/replay_soak.log
