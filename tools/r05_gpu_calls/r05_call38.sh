#!/bin/bash
# round-5 GPU call 38: timing-only bound (WRONG results): K3 if no wave ran more than 4 / 6 / 8 trips of the neighbour scan
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c38
mkdir -p $O
export TMPDIR=/tmp
for v in base cap4 cap6 cap8 base; do
  if [ $v = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  (cd /tmp && rm -rf /tmp/cold_$v && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe_$v.log 2>&1)
  python3 tools/save_stats.py /tmp/cold_$v $O/cold_kernel_stats_$v.csv
  echo "== $v"; grep "icp_linearize" $O/cold_kernel_stats_$v.csv | cut -c1-110
done
unset MH_LIB_OVERRIDE
