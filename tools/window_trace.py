#!/usr/bin/env python3
"""The sliding-window workload for rocprofv3 --kernel-trace --stats: 5 factors x 24 576 points against the configs[1]
map, linearized (a) by mh_icp_linearize_batch and (b) one call at a time, cold and warm.  MH_WINDOW=batch|single|both."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, synth

mode = os.environ.get("MH_WINDOW", "both")
rooms = os.environ.get("MH_ROOMS", "2x5")
nx, ny = (int(v) for v in rooms.split("x"))
ctx = capi.Context(0)
gm = capi.VoxelMap(ctx)
for _, _, xyz in synth.make_map_rooms(nx, ny):
    gm.insert(xyz)
pts, _ = synth.make_scan(128)
R, t = synth.query_pose()
cfg = capi.make_reg_config(**synth.enwide_config())
n, per = 5, 24576
fs = [capi.ICPFactor(ctx, gm, np.ascontiguousarray(pts[i::n][:per]), cfg) for i in range(n)]
Rs, ts = [R] * n, [t + np.array([0.002, -0.001, 0.0005]) * i for i in range(n)]
for rep in range(30):
    cold = rep % 2 == 0
    if cold:
        for f in fs:
            f.reset()
    if mode in ("batch", "both"):
        capi.linearize_batch(fs, Rs, ts)
    if cold:
        for f in fs:
            f.reset()
    if mode in ("single", "both"):
        for i, f in enumerate(fs):
            f.linearize(Rs[i], ts[i])
print("done")
for f in fs:
    f.destroy()
gm.release()
ctx.close()  # (MH_WAIT_TRACE=1 prints the host side's shares at the shutdown)
