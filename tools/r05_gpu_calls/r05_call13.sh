#!/bin/bash
# round-5 GPU call 13: launch cost by API; small-class workgroup size (timing only); wait trace on cold 131k calls only
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c13
mkdir -p $O
./mimosa_amd/lib/variants/launch_cost > $O/launch_cost.txt 2>&1; cat $O/launch_cost.txt
cd /tmp && export TMPDIR=/tmp
for v in new tpb128 tpb64; do
  if [ $v = new ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  COLD_PROBE_CALLS=80 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_$v.log 2>&1
  python3 $R/tools/save_stats.py /tmp/cold_$v $R/$O/cold_stats_$v.csv
  echo == $v; grep "icp_" $R/$O/cold_stats_$v.csv | cut -c1-120; tail -1 $R/$O/cold_$v.log | cut -c1-200
done
unset MH_LIB_OVERRIDE
cd $R
MH_WAIT_TRACE=1 COLD_PROBE_CALLS=200 timeout 300 python tools/k3_cold_probe.py > $O/cold_trace.json 2> $O/cold_trace.err; grep MH_WAIT $O/cold_trace.err
