#!/usr/bin/env python3
"""Run the device scan front end a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, synth
ctx = capi.Context(0)
raw, aux = synth.make_raw_scan(128)
sc = capi.Scan(ctx)
col = {int(t): c for c, t in enumerate(aux["unique_ns"])}
I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
for _ in range(20):
    sc.prepare_input(raw, capi.make_input_config())
    uns = sc.unique_ns()
    sc.deskew(np.stack([aux["Rt12"][col[int(u)]] for u in uns]))
    sc.preprocess_geometric(I3, z3, 0.5, 20, 0.15)
if len(sys.argv) > 1 and sys.argv[1] == "dense":  # every point in the geometric subset
    for _ in range(20):
        sc.prepare_input(raw, capi.make_input_config(point_skip_divisor=1))
        sc.deskew(np.stack([aux["Rt12"][col[int(u)]] for u in sc.unique_ns()]))
        sc.preprocess_geometric(I3, z3, 0.5, 20, 0.15)
print("done")
