"""Subprocess body of tests/test_gpu_dist.py: torch.distributed ("nccl" = RCCL) + the sharding layer
driving the HIP backend, world size 1 on cuda:0.  Prints "OK" on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (torch initialises the device before libmimosa_hip does)
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

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
sh.scatter_scan(scan, R, t, device="cuda")
got = sh.linearize(R, t, device="cuda")
M = ref_cpu.Map()
for c in np.array_split(map_xyz, 3):
    M.insert(c)
ref = ref_cpu.ICP(M, scan, ref_cpu.make_config(**cfg)).linearize(R, t)
assert np.array_equal(got["status_hist"], ref["status_hist"])
assert rel(got["H_ss"], ref["H_ss"]) <= 1e-5 and rel(got["b_s"], ref["b_s"]) <= 1e-5
assert rel(got["loc_trans_comp"], ref["loc_trans_comp"]) <= 1e-5
assert rel(got["loc_rot_comp"], ref["loc_rot_comp"]) <= 1e-5
dist.destroy_process_group()
print("OK")
