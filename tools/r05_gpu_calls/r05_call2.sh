#!/bin/bash
# round-5 GPU call 2: parity suites on the reworked K3 / K4, A/B against the round-4 library on the same box, timelines
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c2
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_overlap.py tests/test_gpu_batch.py tests/test_gpu_configs1.py tests/test_gpu_reported_basis.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in base new; do
  if [ $v = base ]; then export MH_LIB_OVERRIDE=$PWD/mimosa_amd/lib/variants/base.so; else unset MH_LIB_OVERRIDE; fi
  timeout 300 python tools/k3_time.py > $O/k3_time_$v.json 2> $O/k3_time_$v.err
  SYNC_PROBE_CALLS=100 timeout 600 python tools/sync_probe.py > $O/sync_probe_$v.json 2> $O/sync_probe_$v.err
done
unset MH_LIB_OVERRIDE
timeout 300 python tools/timeline.py > $O/timeline_131k.txt 2>&1
MH_TL_POINTS=24576 timeout 300 python tools/timeline.py > $O/timeline_24k.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base new; do
  if [ $v = base ]; then export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/base.so; else unset MH_LIB_OVERRIDE; fi
  SYNC_PROBE_CALLS=40 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python $R/tools/sync_probe.py > $R/$O/trace_$v.log 2>&1
  for f in $(ls /tmp/tr_$v/*/*_kernel_stats.csv 2>/dev/null); do cp $f $R/$O/kernel_stats_$v.csv; done
done
ls -la $R/$O
