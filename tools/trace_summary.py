#!/usr/bin/env python3
"""Per-kernel summary (name, grid, calls, avg/min/max us) of the newest rocprofv3 kernel trace under a directory."""
import collections, csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel|__amd\w+)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:40]
    acc[(n, r.get("Grid_Size_X"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(k[0], k[1], len(v), round(sum(v) / len(v) / 1e3, 2), min(v) / 1e3, max(v) / 1e3)
