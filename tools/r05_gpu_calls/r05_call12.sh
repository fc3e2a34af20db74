#!/bin/bash
# round-5 GPU call 12: scan pipeline depth of the 256-thread class (cold 24 576-point calls), host-side breakdown of the synchronous call
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in new pipe6 pipe8; do
  if [ $v = new ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  COLD_PROBE_CALLS=80 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_$v.log 2>&1
  python3 $R/tools/save_stats.py /tmp/cold_$v $R/$O/cold_stats_$v.csv
  echo == $v; grep icp_ $R/$O/cold_stats_$v.csv | cut -d, -f1-4
done
unset MH_LIB_OVERRIDE
cd $R
MH_WAIT_TRACE=1 SYNC_PROBE_CALLS=200 timeout 600 python tools/sync_probe.py > $O/sync_trace.json 2> $O/sync_trace.err
grep MH_WAIT_TRACE $O/sync_trace.err
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_shard_native.py -q -m gpu > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
