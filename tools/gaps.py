#!/usr/bin/env python3
"""Kernel timeline of the pipelined bench from a rocprofv3 --kernel-trace CSV: durations and the idle
gaps between consecutive kernels on the stream (K3 -> K4 -> next K3)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
rows = [r for r in rows if 'icp_' in r[2]]
rows = rows[len(rows) // 4:]  # skip warm-up
k3 = [(s, e) for s, e, n in rows if 'linearize' in n]
k4 = [(s, e) for s, e, n in rows if 'localizability' in n]
d3 = np.array([e - s for s, e in k3]) / 1e3
d4 = np.array([e - s for s, e in k4]) / 1e3
print(f"K3 n={len(d3)} mean {d3.mean():.2f} us p50 {np.median(d3):.2f} p95 {np.percentile(d3,95):.2f}")
print(f"K4 n={len(d4)} mean {d4.mean():.2f} us p50 {np.median(d4):.2f} p95 {np.percentile(d4,95):.2f}")
g34, g43 = [], []
for i in range(len(rows) - 1):
    (s0, e0, n0), (s1, e1, n1) = rows[i], rows[i + 1]
    g = (s1 - e0) / 1e3
    if 'linearize' in n0 and 'localizability' in n1: g34.append(g)
    if 'localizability' in n0 and 'linearize' in n1: g43.append(g)
g34, g43 = np.array(g34), np.array(g43)
print(f"gap K3->K4 mean {g34.mean():.2f} us p50 {np.median(g34):.2f}; gap K4->next K3 mean {g43.mean():.2f} p50 {np.median(g43):.2f} p95 {np.percentile(g43,95):.2f}")
per = np.diff([s for s, e in k3]) / 1e3
print(f"K3 start-to-start: p50 {np.median(per):.2f} us mean {per.mean():.2f}")
