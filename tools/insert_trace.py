#!/usr/bin/env python3
"""Keyframe map inserts (32 768 points into the configs[1] map) for rocprofv3 --kernel-trace --stats."""
import os, sys
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
q = synth.points_xyz(pts).astype(np.float64)[::4] @ R.T + t
for k in range(12):
    m2 = gmap.copy()
    m2.insert((q + 0.01 * k).astype(np.float32))
    m2.release()
print("done")
