"""bench.py side leg (tools/benchlegs): synchronous per-call latency with and without the component pass, warm re-linearization

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
    # synchronous per-call latency (result on the host before the next call), events off
    for c in ctxs:
        c.set_profiling(False)
    from .env import c_sync_ns
    n_lat = 0 if args.profile_mode else min(50, max(10, args.steps // 4))
    lat = c_sync_ns(ctx, factor.h, R, t, _g, _out, n_lat + 3)[3:] if n_lat else []
    lat_ms = float(np.median(lat) * 1e-6) if len(lat) else float("nan")
    factor.set_components(False)  # the same call with the component pass switched off (K3 alone publishes the result)
    lat_nc = c_sync_ns(ctx, factor.h, R, t, _g, _out, n_lat + 3)[3:] if n_lat else []
    factor.set_components(True)
    lat_nc_ms = float(np.median(lat_nc) * 1e-6) if len(lat_nc) else float("nan")
    lat_py = []  # the same call through ctypes, arguments marshalled once: what the Python binding's caller pays at best
    for _ in range(n_lat):
        factor.reset()
        ctx.synchronize()
        a = time.perf_counter()
        raw_linearize()
        lat_py.append(time.perf_counter() - a)
    # The second caller of the path: GTSAM re-linearization (src/graph/manager.cpp:585-588).  The pose moved
    # < min_dist/4, so every point takes the data-association cache branch (geometric_factor.hpp:308-317):
    # no k-NN, cached plane, residual + Jacobian + reduction only.  Wall time with events off, kernel time in a
    # second loop with events on.
    relin_k3, relin_wall = [], []
    if not args.profile_mode:
        factor.reset()
        factor.linearize(R, t)
    for i in range(0 if args.profile_mode else 30):
        dt = np.array([1e-3, -5e-4, 2e-4]) * ((i % 3) - 1)
        E._t[:] = t + dt
        ctx.synchronize()
        a = time.perf_counter()
        raw_linearize()
        relin_wall.append(time.perf_counter() - a)
    E._t[:] = t
    ctx.set_profiling(1)
    for i in range(0 if args.profile_mode else 30):
        dt = np.array([1e-3, -5e-4, 2e-4]) * ((i % 3) - 1)
        rr = factor.linearize(R, t + dt)
        relin_k3.append(rr["gpu_ms_linearize"])
        assert rr["n_knn"] == 0, "re-linearization leg ran k-NN"
    ctx.set_profiling(False)
    return {
        "sync_latency_ms": round(lat_ms, 4),
        "value_sync": round(n_pts / (lat_ms * 1e-3) / 1e6, 2),
        "sync_latency_without_components_ms": round(lat_nc_ms, 4),
        "value_sync_without_components": round(n_pts / (lat_nc_ms * 1e-3) / 1e6, 2),
        "sync_latency_note": "median wall time of mh_icp_linearize calls made from C (tools/micro/sync_caller.c: clock_gettime around the call, "
                             "association state reset and the stream drained before each): what ICPFactor::linearize costs its C++ caller",
        "sync_latency_through_ctypes_ms": round(float(np.median(lat_py)) * 1e3, 4) if lat_py else None,
        "relinearize": {"what": "warm ICPFactor::linearize (all points hit the data-association cache, no k-NN)",
                        "kernel_ms": round(float(np.median(relin_k3)), 5) if relin_k3 else None,
                        "sync_latency_ms": round(float(np.median(relin_wall) * 1e3), 4) if relin_wall else None,
                        "value_sync": round(n_pts / float(np.median(relin_wall)) / 1e6, 1) if relin_wall else None},
    }
