#!/bin/bash
# round-5 GPU call 34: resident photometric preprocess finished by a completion number instead of a stream synchronisation
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c34
mkdir -p $O
export TMPDIR=/tmp
MH_FUZZ_EXTRA=100 timeout 1200 python -m pytest tests/test_gpu_photo.py tests/test_gpu_photo_fuzz.py tests/test_replay.py tests/test_scan_frontend.py -q -m gpu -x > $O/pytest_photo.log 2>&1; echo "rc $?" >> $O/pytest_photo.log; tail -n 3 $O/pytest_photo.log
for k in 1 2 3; do timeout 300 python tools/photo_resident_time.py 2> $O/photo_resident.err | tee -a $O/photo_resident.jsonl; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 600 python tools/replay_soak.py 4 > $O/replay_soak.log 2>&1; tail -n 1 $O/replay_soak.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "rc $?" >> $O/pytest_full.log; tail -n 3 $O/pytest_full.log
