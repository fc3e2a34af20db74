#!/bin/bash
# round-5 GPU call 37: the final build — full suite, the same under the allocation checker with the fuzz extras, replay soak, smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c37
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/full.log 2>&1; echo "rc $?" >> $O/full.log; tail -n 2 $O/full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
export MH_ALLOC_CHECK=1
MH_FUZZ_EXTRA=300 timeout 2400 python -m pytest tests -q -m gpu > $O/full_checked.log 2>&1; echo "rc $?" >> $O/full_checked.log; tail -n 2 $O/full_checked.log
timeout 1200 python tools/replay_soak.py 8 > $O/replay_soak.log 2>&1; tail -n 1 $O/replay_soak.log
