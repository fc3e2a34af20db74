#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c3
mkdir -p $O
./mimosa_amd/lib/variants/anyorder_test > $O/anyorder.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
MH_OVERLAP=0 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_nooverlap.json 2> $O/bench_nooverlap.err
