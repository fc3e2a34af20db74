#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command's timed region (K3 / K4 average durations).
# usage (GPU box): tools/k3_prof.sh <tag>   -> gpurun_out/k3prof_<tag>.csv (+ prints the icp_ rows)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-x}
OUT=$R/gpurun_out/k3prof_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --steps 200 --warmup 20 --profile-mode --no-measure-traffic > $OUT/log 2>&1
f=$(ls $OUT/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/k3prof_$tag.csv && grep -E "Name|icp_" $f | cut -c1-220
rm -rf $OUT
