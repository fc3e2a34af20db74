"""What does the barrier of bench.py's timed block cost under torch.distributed.run?  Times, per block of 64 pipelined cold
linearizes: the loop itself, ctx.synchronize(), dist.barrier(), an all-reduce of a resident 1-element tensor, torch.cuda.synchronize().
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/torchrun_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from mimosa_amd import capi, synth

lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
ctx = capi.Context(lr)
room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                     max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
for xyz in room_clouds:
    gmap.insert(xyz)
f = capi.ICPFactor(ctx, gmap, pts, capi.make_reg_config(**cfgd))
f.linearize(R, t)
bt = torch.zeros(1, device="cuda")
if os.environ.get("PROBE_PROF"):
    ctx.set_profiling(int(os.environ["PROBE_PROF"]))  # HIP events around every n-th call, as bench.py's timed region has them
rows = []
for blk in range(8):
    ctx.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        f.reset(); f.linearize_async(R, t)
    f.wait()
    t1 = time.perf_counter(); ctx.synchronize()
    t2 = time.perf_counter(); dist.barrier()
    t3 = time.perf_counter(); torch.cuda.synchronize()
    t4 = time.perf_counter(); dist.all_reduce(bt)
    t5 = time.perf_counter(); torch.cuda.synchronize()
    t6 = time.perf_counter()
    rows.append([round((b - a) * 1e6, 1) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))])
if dist.get_rank() == 0:
    print(json.dumps({"us": "loop64, ctx.synchronize, dist.barrier, cuda.synchronize, all_reduce(resident 1 elem), cuda.synchronize", "blocks": rows}))
dist.destroy_process_group()
