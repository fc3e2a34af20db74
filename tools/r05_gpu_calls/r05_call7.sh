#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c7
mkdir -p $O
MH_LIB_OVERRIDE=$PWD/mimosa_amd/lib/variants/nojobs.so timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_configs1.py -q -m gpu > $O/pytest_nojobs.log 2>&1; echo "rc $?" >> $O/pytest_nojobs.log
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_configs1.py tests/test_gpu_fuzz.py -q -m gpu > $O/pytest_jobs.log 2>&1; echo "rc $?" >> $O/pytest_jobs.log
tail -4 $O/pytest_nojobs.log $O/pytest_jobs.log
