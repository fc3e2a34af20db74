"""Subprocess body of tests/test_gpu_dist.py::test_sharded_configs2_size_world2 — BASELINE configs[2] at full size: the
131 072-point scan against the 10 x 10-room map (~50 M points) hash-sharded over TWO ranks (gloo carries the collectives:
RCCL does not allow two ranks on one device and the test box has one GPU; both ranks compute on cuda:0 with
device-resident buffers).  The sharded factor must equal the UNSHARDED HIP factor on the full map (which
tests/test_gpu_configs1.py pins against the oracle at the configs[1] size).  Prints "OK <rank>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)

from mimosa_amd import capi, dist as mdist, synth  # noqa: E402
from parity import rel  # noqa: E402

ctx = capi.Context(0)
cfg = synth.enwide_config()
rc = capi.make_reg_config(**cfg)
kw = dict(leaf=cfg["target_ivox_map_leaf_size"], min_dist=cfg["target_ivox_map_min_dist_in_voxel"], max_pts=synth.MAX_PTS_PER_VOXEL,
          mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
sh = mdist.ShardedICPDevice(dist.group.WORLD, ctx, cfg["target_ivox_map_leaf_size"], rc, torch.device("cuda", 0))
full = capi.VoxelMap(ctx, **kw) if rank == 0 else None


def rooms():  # identical on every rank; rank 0 also builds the unsharded map
    for _, _, xyz in synth.make_map_rooms(10, 10):
        if full is not None:
            full.insert(xyz)
        yield xyz


sh.build_map(rooms(), **kw)
scan, aux = synth.make_scan(128, seed=synth.BASE_SEED + 1)
R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
sh.set_scan(np.array_split(scan, world)[rank])
stats = sh.map.stats()
tot = torch.tensor([float(stats["n_points"])], dtype=torch.float64)
dist.all_reduce(tot)
poses = [(R, t), (R @ synth.so3_exp(np.array([0.0, 0.0, 0.01])), t + np.array([0.05, -0.04, 0.01]))]
ref = capi.ICPFactor(ctx, full, scan, rc) if rank == 0 else None
for Rk, tk in poses:
    got = sh.linearize(Rk, tk)
    if rank == 0:
        want = ref.linearize(Rk, tk)
        assert np.array_equal(got["status_hist"], want["status_hist"]), (got["status_hist"], want["status_hist"])
        assert want["status_hist"][8] > 50000
        assert rel(got["H_ss"], want["H_ss"]) <= 1e-9 and rel(got["b_s"], want["b_s"]) <= 1e-9
        assert abs(got["f"] - want["f"]) <= 1e-9 * abs(want["f"])
        assert rel(got["loc_trans_comp"], want["loc_trans_comp"]) <= 1e-9 and rel(got["loc_rot_comp"], want["loc_rot_comp"]) <= 1e-9
if rank == 0:
    fs = full.stats()
    assert fs["n_points"] > 45_000_000, fs                                   # configs[2]: a ~50 M-point map
    # shards + one-voxel halos of 8^3-voxel blocks: more than the map (surfaces: up to (10/8)^2 each), less than two copies
    assert fs["n_points"] < tot.item() < 2.0 * fs["n_points"], (fs["n_points"], tot.item())
    assert stats["n_points"] < 0.95 * fs["n_points"], (stats["n_points"], fs["n_points"])   # rank 0 holds a shard, not the map
    print("map", fs["n_points"], "stored in shards", int(tot.item()), "rank 0", stats["n_points"])
sh.close()
dist.destroy_process_group()
print("OK", rank)
