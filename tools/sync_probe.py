#!/usr/bin/env python3
"""Synchronous mh_icp_linearize latency, calls made from C (tools/micro/sync_caller.c: clock_gettime around the foreign call):
with and without the component pass, at 131 072 and 24 576 points on the configs[1] world.
Prints one JSON line.  Under `rocprofv3 --hip-trace --kernel-trace` the same calls give the API / kernel breakdown."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimosa_amd import capi, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlegs.env import c_sync_ns  # noqa: E402

N = int(os.environ.get("SYNC_PROBE_CALLS", "200"))
room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
rc = capi.make_reg_config(**cfgd)
out = {}
for mode, env in (("two_launch", {}),):
    os.environ.update(env)
    ctx = capi.Context(0)
    gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                         max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
    for xyz in room_clouds:
        gmap.insert(xyz)
    for name, cloud in (("131k", pts), ("24k", np.ascontiguousarray(pts[::5][:24576]))):
        f = capi.ICPFactor(ctx, gmap, cloud, rc)
        res = capi.IcpResult()
        Rc, tc, gc = np.ascontiguousarray(R, np.float64), np.ascontiguousarray(t, np.float64), np.array([0.0, 0.0, -1.0])
        for comp in (1, 0):
            f.set_components(bool(comp))
            cold = c_sync_ns(ctx, f.h, Rc, tc, gc, res, N + 20, reset=True)[20:]
            warm = c_sync_ns(ctx, f.h, Rc, tc, gc, res, N // 2, reset=False)
            q = lambda v, p: round(float(np.percentile(v, p)) / 1e3, 2)
            out[f"{mode}.{name}.{'comp' if comp else 'nocomp'}"] = {"cold_p10": q(cold, 10), "cold_p50": q(cold, 50), "cold_p90": q(cold, 90), "warm_p50": q(warm, 50)}
        out[f"{mode}.{name}.H00"] = float(res.as_dict()["H_ss"][0][0])
        out[f"{mode}.{name}.loc"] = [float(x) for x in res.as_dict()["loc_trans_final"]]
        f.destroy()
    gmap.release()
    ctx.close()
print(json.dumps(out))
