"""Subprocess body of tests/test_gpu_shard_native.py::test_native_world1_over_rccl_full_protocol: one rank, librccl resolved by
the library itself (no torch in this process), the FULL exchange protocol forced — ncclAllToAll of the segments to itself,
ncclAllReduce of the sums — against the unsharded oracle.  Prints "OK"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from mimosa_amd import capi  # noqa: E402
import shard_native_common as C  # noqa: E402

ctx = capi.Context(0)
comm = capi.ShardComm.rccl(ctx, capi.ShardComm.unique_id(), 1, 0)
assert comm.backend == "rccl" and comm.world == 1
case = C.default_case()
refs, n_map = C.oracle_results(case)
split = [np.arange(len(case["scan"]))]
res = C.run_rank(comm, ctx, case, refs, split, force=True)
assert res["stats"]["collective"] == 1 and res["stats"]["collectives_last"] == 3, res["stats"]
assert res["map_points"] == n_map
res = C.run_rank(comm, ctx, C.default_case(dict(reg_4_dof=1)), C.oracle_results(C.default_case(dict(reg_4_dof=1)))[0], split, force=True)
comm.destroy()
ctx.close()
print("OK")
