"""bench.py side leg (tools/benchlegs): PCIe-inclusive figure: factor creation + first linearize

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
    # PCIe-inclusive figure: the boundary hands over HOST buffers, so a scan costs a factor creation
    # (4 MiB upload + pack + Morton sort) before its first linearize.  Reported, never the headline.
    cre = []
    for _ in range(0 if args.profile_mode else 5):
        ctx.synchronize()
        a = time.perf_counter()
        f2 = capi.ICPFactor(ctx, gmap, pts, capi.make_reg_config(**cfgd))
        f2.linearize(R, t)
        cre.append(time.perf_counter() - a)
        f2.destroy()
    create_plus_lin_ms = float(np.median(cre) * 1e3) if cre else float("nan")
    return {"value_pcie_inclusive": round(n_pts / (create_plus_lin_ms * 1e-3) / 1e6, 2), "create_plus_linearize_ms": round(create_plus_lin_ms, 4)}
