#!/usr/bin/env python3
"""Kernel-tuning probe: K3 / K4 HIP-event times and the synchronous call latency of a cold linearize on the configs[1]
world, for the full 131 072-point scan and the 24 576-point cloud, plus the 5-factor window through the batch entry
point.  Prints one JSON line.  MH_LIB_OVERRIDE selects a variant build (tools/variant.sh)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mimosa_amd import capi, synth

ctx = capi.Context(0)
room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                     max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
for xyz in room_clouds:
    gmap.insert(xyz)
rc = capi.make_reg_config(**cfgd)
out = {"lib": os.environ.get("MH_LIB_OVERRIDE", "default")}
for name, cloud in (("131k", pts), ("24k", np.ascontiguousarray(pts[::5][:24576]))):
    f = capi.ICPFactor(ctx, gmap, cloud, rc)
    ctx.set_profiling(1)
    k3, k4 = [], []
    for _ in range(40):
        f.reset()
        r = f.linearize(R, t)
        k3.append(r["gpu_ms_linearize"]); k4.append(r["gpu_ms_localizability"])
    ctx.set_profiling(False)
    sl = []
    for _ in range(40):
        f.reset(); ctx.synchronize()
        a = time.perf_counter(); f.linearize(R, t); sl.append(time.perf_counter() - a)
    warm = []
    for _ in range(20):
        ctx.synchronize(); a = time.perf_counter(); f.linearize(R, t); warm.append(time.perf_counter() - a)
    out[name] = {"k3_us": round(float(np.mean(k3[8:])) * 1e3, 2), "k4_us": round(float(np.mean(k4[8:])) * 1e3, 2),
                 "sync_us": round(float(np.median(sl)) * 1e6, 1), "warm_sync_us": round(float(np.median(warm)) * 1e6, 1), "H00": float(np.asarray(r["H_ss"]).ravel()[0])}
    f.destroy()
fs = [capi.ICPFactor(ctx, gmap, np.ascontiguousarray(pts[k::5][:24576]), rc) for k in range(5)]
Rs, ts = [R] * 5, [t] * 5
bt = []
for _ in range(30):
    for f in fs:
        f.reset()
    ctx.synchronize(); a = time.perf_counter(); capi.linearize_batch(fs, Rs, ts); bt.append(time.perf_counter() - a)
out["window5_cold_us"] = round(float(np.median(bt)) * 1e6, 1)
print(json.dumps(out))
