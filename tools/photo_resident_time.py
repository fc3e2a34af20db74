#!/usr/bin/env python3
"""Resident Photometric::preprocess (mh_photo_preprocess_scan: clouds already on the device, <= 1024 poses uploaded):
median wall time of raw C-ABI calls, one JSON line.  The same loop as bench.py's photometric.resident block."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, replay as rp  # noqa: E402

ctx = capi.Context(0)
rc1 = rp.ReplayConfig(n_scans=1)
s1 = rp.make_scans(rc1)[0]
scp = capi.Scan(ctx)
ctx.check(ctx.L.mh_scan_keep_raw(scp.h, 1))
G = capi.Photo(ctx, rc1.photo)
tr = []
for it in range(int(os.environ.get("REPS", "40")) + 1):
    scp.prepare_input(s1["raw"], capi.make_input_config())
    Tq = np.ascontiguousarray(s1["frame"]["T_Le_Lt"][np.searchsorted(s1["frame"]["unique_ns"], scp.unique_ns())], np.float64)
    scp.deskew(Tq.astype(np.float32))
    ctx.synchronize()
    a = time.perf_counter_ns()
    ctx.check(ctx.L.mh_photo_preprocess_scan(G.h, scp.h, Tq.ctypes.data_as(C.c_void_p), len(Tq)))
    b = time.perf_counter_ns()
    if it:
        tr.append((b - a) * 1e-3)
print(json.dumps({"preprocess_scan_us_p50": round(float(np.median(tr)), 2), "p10": round(float(np.percentile(tr, 10)), 2),
                  "p90": round(float(np.percentile(tr, 90)), 2), "calls": len(tr)}))
