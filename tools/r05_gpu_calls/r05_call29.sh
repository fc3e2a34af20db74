#!/bin/bash
# round-5 GPU call 29: knobs of the neighbour scan under the strided chunk mapping — quads in flight (MH_PIPE 2 / 3 / 6), how often the
# box-distance pruning is re-applied (MH_PRUNE_TRIPS), 256-thread workgroups at 131 072 points (MH_TPB_SPLIT)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c29
mkdir -p $O
export TMPDIR=/tmp
for v in base pipe2 pipe3 pipe6 pruneall prune14 tpb256 base; do
  if [ $v = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  (cd /tmp && rm -rf /tmp/cold_$v && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cold_$v -- python $R/tools/k3_cold_probe.py > $R/$O/cold_probe_$v.log 2>&1)
  python3 tools/save_stats.py /tmp/cold_$v $O/cold_kernel_stats_$v.csv
  echo "== $v"; grep "icp_l" $O/cold_kernel_stats_$v.csv | cut -c1-110
done
unset MH_LIB_OVERRIDE
