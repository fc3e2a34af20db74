"""Subprocess body of tests/test_gpu_dist.py::test_sharded_layer_on_hip_backend_world2: TWO ranks (gloo for the
collectives, CPU tensors) both driving the HIP backend on cuda:0 — the sharded map, the point routing all-to-all,
the Hessian all-reduce and the two-phase linearize with a real second rank.  (RCCL does not allow two ranks on one
device, and the test box has one GPU.)  Prints "OK <rank>" on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

from mimosa_amd import capi, dist as mdist, synth  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from parity import rel  # noqa: E402

ctx = capi.Context(0)
room = np.array([20.0, 14.0, 3.0])
map_xyz = synth.make_room(4321, 0, 0, room=room)
scan, aux = synth.make_scan(n_rows=32, seed=99, n_cols=128, room=room, sensor_local=np.array([9.3, 6.6, 1.2]))
R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
cfg = synth.enwide_config()
sh = mdist.ShardedICP(dist.group.WORLD, lambda: capi.VoxelMap(ctx),
                      lambda m, pts: capi.ICPFactor(ctx, m, pts, capi.make_reg_config(**cfg)), leaf=0.5)
sh.build_map(np.array_split(map_xyz, 3))
# each rank holds half of the scan before routing, as if it had come from two sensors / two halves of a sweep
mine = scan[rank::world]
sh.scatter_scan(mine, R, t, device="cpu")
got = sh.linearize(R, t, device="cpu")
M = ref_cpu.Map()
for c in np.array_split(map_xyz, 3):
    M.insert(c)
ref = ref_cpu.ICP(M, scan, ref_cpu.make_config(**cfg)).linearize(R, t)
assert np.array_equal(got["status_hist"], ref["status_hist"]), (got["status_hist"], ref["status_hist"])
assert rel(got["H_ss"], ref["H_ss"]) <= 1e-5 and rel(got["b_s"], ref["b_s"]) <= 1e-5
assert abs(got["f"] - ref["f"]) <= 1e-5 * ref["f"]
assert rel(got["loc_trans_comp"], ref["loc_trans_comp"]) <= 1e-5
assert rel(got["loc_rot_comp"], ref["loc_rot_comp"]) <= 1e-5
st = sh.map.stats()
assert 0 < st["n_points"] < M.num_points  # a real shard (with halo), not the whole map
dist.destroy_process_group()
print("OK", rank)
