#!/usr/bin/env python3
"""Latency probe of the native map-sharded factor (mh_shard_*) on the configs[1] world with ONE rank: the plain factor, the
sharded factor without collectives (what world 1 runs), and the full exchange protocol forced over RCCL (self all-to-all +
all-reduces) — cold (association state reset), warm (same pose), with and without the component pass, and a walking pose.
Prints one JSON line."""
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
comm = capi.ShardComm.rccl(ctx, capi.ShardComm.unique_id(), 1, 0)
out = {"backend": comm.backend}


def med_us(fn, n=40, pre=None):
    v = []
    for _ in range(n):
        if pre:
            pre()
        ctx.synchronize()
        a = time.perf_counter(); fn(); v.append(time.perf_counter() - a)
    return round(float(np.median(v[5:])) * 1e6, 1)


plain = capi.ICPFactor(ctx, gmap, pts, rc)
plain.linearize(R, t)
out["plain"] = {"cold_us": med_us(lambda: plain.linearize(R, t), pre=plain.reset), "warm_us": med_us(lambda: plain.linearize(R, t))}
plain.set_components(False)
out["plain"].update(cold_nocomp_us=med_us(lambda: plain.linearize(R, t), pre=plain.reset), warm_nocomp_us=med_us(lambda: plain.linearize(R, t)))
plain.destroy()
for name, force in (("sharded_world1", False), ("sharded_world1_full_protocol", True)):
    a0 = time.perf_counter()
    f = capi.ShardedICPFactor(ctx, comm, gmap, pts, rc, force_collectives=force)
    create_us = (time.perf_counter() - a0) * 1e6
    a0 = time.perf_counter()
    first = f.linearize(R, t)
    first_us = (time.perf_counter() - a0) * 1e6
    d = {"create_us": round(create_us, 1), "first_us": round(first_us, 1), "cold_us": med_us(lambda: f.linearize(R, t), pre=f.reset), "warm_us": med_us(lambda: f.linearize(R, t))}
    f.set_components(False)
    d.update(cold_nocomp_us=med_us(lambda: f.linearize(R, t), pre=f.reset), warm_nocomp_us=med_us(lambda: f.linearize(R, t)))
    # walking pose: 1 cm / 1 mrad per call
    k = [0]
    def walk():
        k[0] += 1
        return f.linearize(R @ synth.so3_exp(np.array([0, 0, 0.001 * k[0]])), t + np.array([0.01, 0.004, 0.0]) * k[0])
    d["walk_nocomp_us"] = med_us(walk, n=30)
    d["stats"] = f.stats()
    d["H00"] = float(first["H_ss"][0, 0])
    out[name] = d
    f.destroy()
comm.destroy()
print(json.dumps(out))
