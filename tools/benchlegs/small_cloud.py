"""bench.py side leg (tools/benchlegs): the 24 576-point cloud the reference really feeds the factor

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
    # the cloud size the reference actually feeds the factor (10-25 k points after the down-sampler)
    small = None
    if not args.profile_mode and world == 1:
        ps = np.ascontiguousarray(pts[::5][:24576])
        fs = capi.ICPFactor(ctx, gmap, ps, capi.make_reg_config(**cfgd))
        ctx.set_profiling(1)
        k3s, k4s, cqs = [], [], 0.0
        for _ in range(30):
            fs.reset()
            rs = fs.linearize(R, t)
            k3s.append(rs["gpu_ms_linearize"])
            k4s.append(rs["gpu_ms_localizability"])
            cqs = float(rs["mean_candidates"])
        ctx.set_profiling(False)
        sl, _ts = [], np.ascontiguousarray(t, np.float64)
        _lin, _raw = ctx.L.mh_icp_linearize, (fs.h, _R.ctypes.data_as(C.c_void_p), _ts.ctypes.data_as(C.c_void_p), None, None,
                                              _g.ctypes.data_as(C.c_void_p), C.byref(_out))
        for _ in range(30):
            fs.reset()
            ctx.synchronize()
            a = time.perf_counter()
            rc = _lin(*_raw)
            sl.append(time.perf_counter() - a)
        bs = 384.0 + 16.0 * cqs
        small = {"points": int(len(ps)), "kernel_ms_avg": round(float(np.mean(k3s[5:])), 5), "localizability_kernel_ms_avg": round(float(np.mean(k4s[5:])), 5),
                 "sync_latency_ms": round(float(np.median(sl)) * 1e3, 4), "value_sync": round(len(ps) / float(np.median(sl)) / 1e6, 1),
                 "achieved": round(len(ps) * bs / (float(np.mean(k3s[5:])) * 1e-3) / 1e9, 1),
                 "frac": round(len(ps) * bs / (float(np.mean(k3s[5:])) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        fs.destroy()

    return {"small_cloud": small}
