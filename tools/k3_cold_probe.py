#!/usr/bin/env python3
"""Cold synchronous linearize calls only (association state reset before each), at 131 072 and 24 576 points on the configs[1] world:
under `rocprofv3 --kernel-trace --stats` the averages of icp_linearize_kernel<.., 512, ..> / <.., 256, ..> and of the two
icp_localizability_kernel instantiations are the cold-call kernel times at the two sizes.  Prints the wall-clock medians."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimosa_amd import capi, synth  # noqa: E402

N = int(os.environ.get("COLD_PROBE_CALLS", "120"))
ctx = capi.Context(0)
room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                     max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
for xyz in room_clouds:
    gmap.insert(xyz)
out = {}
for name, cloud in (("131k", pts), ("24k", np.ascontiguousarray(pts[::5][:24576]))):
    f = capi.ICPFactor(ctx, gmap, cloud, capi.make_reg_config(**cfgd))
    ts = []
    for i in range(N):
        f.reset()
        ctx.synchronize()
        a = time.perf_counter()
        r = f.linearize(R, t)
        ts.append(time.perf_counter() - a)
    out[name] = {"calls": N, "sync_us_python_binding_p50": round(float(np.median(ts[10:])) * 1e6, 1), "mean_scanned": round(float(r["mean_scanned"]), 2),
                 "exact_fallback": int(r["n_exact_fallback"])}
    f.destroy()
print(json.dumps(out))
