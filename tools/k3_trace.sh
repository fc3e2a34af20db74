#!/bin/bash
# rocprofv3 kernel trace of the default bench command's pipelined loop: start / end of consecutive K3 / K4 dispatches (us, relative),
# i.e. whether K4 of one call really runs beside K3 of the next.  usage (GPU box): tools/k3_trace.sh <tag> -> gpurun_out/k3trace_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-x}
OUT=$R/gpurun_out/k3trace_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --steps ${STEPS:-200} --warmup ${WARM:-20} --profile-mode --no-measure-traffic > $OUT/log 2>&1
f=$(ls $OUT/*/*_kernel_trace.csv 2>/dev/null | head -1)
python - "$f" > $R/gpurun_out/k3trace_$tag.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "icp_l" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2 if "FROM0" not in __import__("os").environ else max(0, len(rows) - int(__import__("os").environ.get("ROWS", "16")))
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + int(__import__("os").environ.get("ROWS", "16"))]:
    name = "K3" if "linearize" in r["Kernel_Name"] else "K4"
    print(name, "queue", r.get("Queue_Id"), "start %.1f end %.1f dur %.1f" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
k3 = [r for r in rows if "linearize" in r["Kernel_Name"]]
d = sorted((int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(k3[20:-1], k3[21:]))
print("K3 start-to-start us: median %.1f p10 %.1f p90 %.1f" % (d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10]))
PY
cat $R/gpurun_out/k3trace_$tag.txt
rm -rf $OUT
