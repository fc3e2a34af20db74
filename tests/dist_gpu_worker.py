"""Subprocess body of tests/test_gpu_dist.py: torch.distributed ("nccl" = RCCL) + the sharding layer on the HIP backend,
world size 1 on cuda:0 (the only size the one-GPU test box offers to RCCL), the libmimosa_hip context bound to torch's
current stream so kernels, collectives and tensor ops are ordered without host synchronisation.  Prints "OK" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (torch initialises the device before libmimosa_hip does)
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

from mimosa_amd import dist as mdist  # noqa: E402
import dist_gpu_common  # noqa: E402

ctx = mdist.context_on_torch_stream(0)
dist_gpu_common.run(dist, ctx, torch.device("cuda", 0))
dist_gpu_common.run(dist, ctx, torch.device("cuda", 0), dict(reg_4_dof=1))   # the epilogue runs once, on the global sums
dist.destroy_process_group()
print("OK")
