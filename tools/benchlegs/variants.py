"""bench.py side leg (tools/benchlegs): the pipelined pass without event records / without the component pass

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
    # the pipelined form (no event records): <= 64 calls of the factor in flight
    barrier()
    a = time.perf_counter()
    if not args.profile_mode:
        run_steps(args.steps)
    barrier()
    elapsed_noev = max(time.perf_counter() - a, 1e-9)
    # the same pipelined pass with the component pass switched off: every step is K3 alone
    for f in factors:
        f.set_components(False)
    barrier()
    a = time.perf_counter()
    if not args.profile_mode:
        run_steps(args.steps)
    barrier()
    elapsed_nocomp = max(time.perf_counter() - a, 1e-9)
    for f in factors:
        f.set_components(True)
    total_pts = n_pts * args.steps * world
    return {"value_pipelined": round(total_pts / elapsed_noev / 1e6, 2), "ms_per_step_pipelined": round(elapsed_noev / max(args.steps, 1) * 1e3, 5),
            "value_pipelined_without_components": round(total_pts / elapsed_nocomp / 1e6, 2),
            "value_pipelined_note": f"<= {INFLIGHT_ICP} cold mh_icp_linearize_async calls of the factor in flight, results collected per burst (K3 and K4 on "
                                    "one stream; rounds 4-5 ran K4 on a side-stream server for this pattern: removed, no caller of the reference pipelines)",
            "_kernel_ms_back_to_back": round(elapsed_nocomp / max(args.steps, 1) * 1e3, 5) if not args.profile_mode else None}
