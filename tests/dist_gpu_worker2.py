"""Subprocess body of tests/test_gpu_dist.py::test_sharded_layer_on_hip_backend_multi_rank: WORLD_SIZE ranks (gloo carries the
collectives: RCCL does not allow two ranks on one device and the test box has one GPU) both driving the HIP backend on
cuda:0 with device-resident buffers.  Prints "OK <rank>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)

from mimosa_amd import capi  # noqa: E402
import dist_gpu_common  # noqa: E402

ctx = capi.Context(0)
moved = dist_gpu_common.run(dist, ctx, torch.device("cuda", 0))
assert moved > 0, "no point changed owner between the poses"
dist_gpu_common.run(dist, ctx, torch.device("cuda", 0), dict(project_on_degneneracy=1, degen_thresh_trans=1e9))
# MH_FUZZ_EXTRA=N: N random configurations on top (a bug hunt on demand)
for seed in range(2 + int(os.environ.get("MH_FUZZ_EXTRA", "0"))):
    dist_gpu_common.run_random(dist, ctx, torch.device("cuda", 0), seed)
dist.destroy_process_group()
print("OK", rank)
