#!/bin/bash
# round-5 GPU call 9: full GPU suite (incl. the configs[2] native sharded tests) on the restored kernel, the refactored bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c9
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 14 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 400 $O/bench.err
