#!/bin/bash
# round-5 GPU call 31: photometric factor kernel, same-box A/B of the three latency cuts (altitude table in LDS | yaw window at once |
# pose index by interpolation guess) against the committed kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=gpurun_out/c31
mkdir -p $O
export TMPDIR=/tmp
for v in base photo_all photo_alt photo_win photo_guess base; do
  if [ $v = base ]; then unset MH_LIB_OVERRIDE; else export MH_LIB_OVERRIDE=$R/mimosa_amd/lib/variants/$v.so; fi
  (cd /tmp && rm -rf /tmp/ptrace && MH_PATCH=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptrace -- python $R/tools/photo_trace.py > $R/$O/photo_trace_$v.log 2>&1)
  python3 tools/save_stats.py /tmp/ptrace $O/photo_trace_kernel_stats_$v.csv; echo "== $v $(grep photo_linearize $O/photo_trace_kernel_stats_$v.csv | cut -c1-90)"
done
unset MH_LIB_OVERRIDE
