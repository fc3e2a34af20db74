"""bench.py side leg (tools/benchlegs): keyframe map update (copy + insert)

Not part of the headline: bench.py's timed region, roofline and cpu_baseline live in bench.py itself.  `run(E)` takes the
shared objects of the run (tools/benchlegs/env.py: Env) and returns the JSON keys it contributes."""
import ctypes as C  # noqa: F401
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .env import HBM_COPY_GBS, HBM_PEAK_GBS, INFLIGHT, INFLIGHT_ICP, ROOT  # noqa: F401


def run(E):
    args, rank, local_rank, world, dist = E.args, E.rank, E.local_rank, E.world, E.dist
    ctx, ctxs, gmap, factor, factors = E.ctx, E.ctxs, E.gmap, E.factor, E.factors
    pts, R, t, cfgd, n_pts, room_clouds = E.pts, E.R, E.t, E.cfgd, E.n_pts, E.room_clouds
    capi, synth, barrier, run_steps, raw_linearize = E.capi, E.synth, E.barrier, E.run_steps, E.raw_linearize
    _R, _g, _out, _all_reduce = E._R, E._g, E._out, E._all_reduce
    # Keyframe map update (Geometric::updateMap, geometric.cpp:427-513): copy the map, insert the scan's geometric
    # subset (every 4th point, world frame).  The map is maintained on the device: copy = device-to-device, insert =
    # the batch over PCIe + the insert kernels (host buffer), or nothing over PCIe (resident scan, see sequence_replay).
    kf_stats = None
    if not args.profile_mode and world == 1:
        sub = pts[::4]
        xyz = synth.points_xyz(sub).astype(np.float64) @ R.T + t
        xyz = np.ascontiguousarray(xyz.astype(np.float32))
        tc, ti = [], []
        for it in range(6):
            ctx.synchronize()
            a0 = time.perf_counter()
            gmap2 = gmap.copy()
            a1 = time.perf_counter()
            gmap2.insert(xyz)
            a2 = time.perf_counter()
            s1 = gmap2.stats()
            gmap2.release()
            if it:
                tc.append(a1 - a0)
                ti.append(a2 - a1)
        kf_stats = {"points": int(len(xyz)), "copy_ms": round(float(np.median(tc)) * 1e3, 3), "insert_ms": round(float(np.median(ti)) * 1e3, 3),
                    "update_ms": round(float(np.median(tc) + np.median(ti)) * 1e3, 3), "bytes_uploaded_per_insert": int(len(xyz) * 12),
                    "map_bytes": int(s1["device_bytes"]), "points_after": int(s1["n_points"]),
                    "note": "host-buffer insert through the Python binding; the map (buckets, block tables, hash, LRU stamps) is built and kept on the device"}
        if not args.no_cpu_baseline:
            from oracle import ref_cpu as _rc
            om = _rc.Map(leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"], max_pts=synth.MAX_PTS_PER_VOXEL,
                         mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
            for xyz_ in room_clouds:
                om.insert(xyz_)
            b0 = time.perf_counter()
            om2 = om.copy()
            b1 = time.perf_counter()
            om2.insert(xyz)
            b2 = time.perf_counter()
            kf_stats["cpu_oracle"] = {"copy_ms": round((b1 - b0) * 1e3, 2), "insert_ms": round((b2 - b1) * 1e3, 2)}
            assert om2.num_points == kf_stats["points_after"], "device and oracle maps disagree after the keyframe insert"

    return {"keyframe_map_update": kf_stats}
