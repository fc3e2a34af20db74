#!/bin/bash
# round-5 GPU call 23: the slow hipLaunchKernel under torch.distributed, seen from inside the runtime (AMD_LOG_LEVEL=4)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c23
mkdir -p $O
AMD_LOG_LEVEL=4 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 tools/torchrun_stall.py > $O/stall.jsonl 2> /tmp/amdlog.txt
ls -la /tmp/amdlog.txt
python3 - <<'PY'
import re, json
# ROCclr log lines: ":<level>:<file>:<line>: <timestamp us> us: [pid:... tid:...] text"
pat = re.compile(r"^:(\d):([\w./]+)\s*:(\d+)\s*:\s*(\d+) us:\s*(?:\[pid:\s*(\d+)\s*tid:\s*(0x[0-9a-f]+)\])?(.*)$")
rows = []
with open("/tmp/amdlog.txt", errors="replace") as f:
    for i, l in enumerate(f):
        m = pat.match(l.rstrip("\n"))
        if m:
            rows.append((int(m.group(4)), m.group(6) or "", m.group(2), m.group(3), m.group(7).strip()[:200], i))
print("log lines parsed:", len(rows))
# per thread: the largest gaps between consecutive lines
by = {}
for r in rows:
    by.setdefault(r[1], []).append(r)
gaps = []
for tid, rs in by.items():
    for a, b in zip(rs, rs[1:]):
        gaps.append((b[0] - a[0], tid, a, b))
gaps.sort(key=lambda g: -g[0])
main_tid = max(by, key=lambda t: sum(1 for r in by[t] if "hipLaunchKernel" in r[4]))  # the thread that launches kernels
print("threads:", {t: len(rs) for t, rs in by.items()}, "main:", main_tid)
mid = [g for g in gaps if g[1] == main_tid and 20000 < g[0] < 75000]
out = []
for g in mid[:6]:
    dt, tid, a, b = g
    others = {}
    for t, rs in by.items():
        if t == tid:
            continue
        inside = [x for x in rs if a[0] <= x[0] <= b[0]]
        if inside:
            others[t] = {"lines": len(inside), "first": [f"{x[0]} {x[2]}:{x[3]} {x[4]}" for x in inside[:6]], "last": [f"{x[0]} {x[2]}:{x[3]} {x[4]}" for x in inside[-4:]]}
    rs = by[tid]
    k = rs.index(a)
    out.append({"gap_us": dt, "tid": tid, "context": [f"{x[0]} {x[2]}:{x[3]} {x[4]}" for x in rs[max(0, k - 14):k + 10]], "other_threads_inside": others})
for o in out:
    print("MAIN-THREAD GAP", o["gap_us"], "us")
    for c in o["context"]:
        print("    ", c[:230])
    for t, v in o["other_threads_inside"].items():
        print("   other thread", t, v["lines"], "lines; first:")
        for c in v["first"]:
            print("        ", c[:200])
        print("      last:")
        for c in v["last"]:
            print("        ", c[:200])
for g in gaps[:6]:
    dt, tid, a, b = g
    # context: 12 lines of that thread before and 6 after
    rs = by[tid]
    k = rs.index(a)
    ctx = [f"{x[0]} {x[2]}:{x[3]} {x[4]}" for x in rs[max(0, k - 12):k + 8]]
    out.append({"gap_us": dt, "tid": tid, "context": ctx})
json.dump(out, open("gpurun_out/c23/runtime_gaps.json", "w"), indent=1)
for o in []:
    print("GAP", o["gap_us"], "us  tid", o["tid"])
    for c in o["context"]:
        print("    ", c[:230])
PY
grep -h "burst_ms" $O/stall.jsonl | cut -c1-200 | head -5
