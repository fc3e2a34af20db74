#!/bin/bash
# Every icp_* dispatch of the default bench command's pipelined loop (rocprofv3 kernel trace): start, end, duration, queue — to see
# WHICH K3 launches are the slow ones (first of a burst? beside a server dispatch?).  usage (GPU box): tools/k3_outliers.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-x}
OUT=$R/gpurun_out/k3out_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --steps 200 --warmup 20 --profile-mode --no-measure-traffic > $OUT/log 2>&1
f=$(ls $OUT/*/*_kernel_trace.csv 2>/dev/null | head -1)
python - "$f" > $R/gpurun_out/k3out_$tag.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "icp_l" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    name = "SRV" if "server" in n else ("K3" if "linearize" in n else "K4")
    print(i, name, "q", r.get("Queue_Id"), "start %.1f end %.1f dur %.1f" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $OUT
