#!/usr/bin/env python3
"""Copy the newest rocprofv3 *_kernel_stats.csv under <dir> to <out.csv> with the kernel names shortened to the function
name (rocPRIM's template names run to kilobytes)."""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"), key=os.path.getmtime)  # gpurun_out/ accumulates: newest, not last by name
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rocprim" in n:
        m = re.search(r"(wrapped_\w+?_config|init_lookback_scan_state_kernel)", n)
        n = "rocprim::" + (m.group(1) if m else "kernel")
    else:
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd\w+)", n)
        n = m.group(1) if m else n[:60]
    w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
