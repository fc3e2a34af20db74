"""bench.py side leg (tools/benchlegs): the smoother window through mh_icp_linearize_batch (relinearize_window)

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
    # The sliding window (src/graph/manager.cpp:585-588: smoother_->update + additional_update_iterations re-linearize
    # every live ICPFactor): 5 factors of ~24 k points (the size the reference's down-sampler feeds the factor, SURVEY
    # F7), one mh_icp_linearize_batch call (one K3 + one K4 launch) vs the same five one call at a time.
    win_stats = None
    if not args.profile_mode and world == 1:
        nwin, per = 5, 24576
        wf = [capi.ICPFactor(ctx, gmap, np.ascontiguousarray(pts[i::nwin][:per]), capi.make_reg_config(**cfgd)) for i in range(nwin)]
        wR = [R for _ in range(nwin)]
        wt = [t + np.array([0.002, -0.001, 0.0005]) * i for i in range(nwin)]
        # raw C-ABI calls with pre-marshalled arguments (what a C++ caller pays; the Python binding's result -> dict
        # conversion costs ~20 us per factor)
        wRa = np.ascontiguousarray(np.stack(wR).reshape(nwin, 9))
        wta = np.ascontiguousarray(np.stack(wt))
        wga = np.ascontiguousarray(np.tile([0.0, 0.0, -1.0], (nwin, 1)))
        whs = (C.c_void_p * nwin)(*[f.h for f in wf])
        wout = (capi.IcpResult * nwin)()
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        def _batch():
            rc = ctx.L.mh_icp_linearize_batch(whs, nwin, vp(wRa), vp(wta), None, None, vp(wga), wout)
            assert rc == 0, rc
        def _one(i):
            rc = ctx.L.mh_icp_linearize(wf[i].h, vp(wRa[i]), vp(wta[i]), None, None, vp(wga[i]), C.byref(wout[i]))
            assert rc == 0, rc
        def _one_by_one():
            for i in range(nwin):
                _one(i)
        def _timed(fn, reps=40, cold=True):
            ts_ = []
            for _ in range(reps):
                if cold:
                    for f in wf:
                        f.reset()
                ctx.synchronize()
                a = time.perf_counter()
                fn()
                ts_.append(time.perf_counter() - a)
            return float(np.median(ts_) * 1e3)
        _batch()
        batch_cold = _timed(_batch)
        single_cold = _timed(_one_by_one)
        one_cold = _timed(lambda: _one(0))
        _batch()
        batch_relin = _timed(_batch, cold=False)
        single_relin = _timed(_one_by_one, cold=False)
        one_relin = _timed(lambda: _one(0), cold=False)
        # the same window with the component pass switched off (mh_icp_set_components(icp, 0)): what the smoother's
        # re-linearizations need — the reference never reads the components of those calls (geometric.cpp:205-214)
        for f in wf:
            f.set_components(False)
        _batch()
        nc_batch_cold = _timed(_batch)
        nc_one_cold = _timed(lambda: _one(0))
        _batch()
        nc_batch_relin = _timed(_batch, cold=False)
        nc_one_relin = _timed(lambda: _one(0), cold=False)
        for f in wf:
            f.set_components(True)
        win_stats = {"factors": nwin, "points_per_factor": per,
                     "batch_cold_ms": round(batch_cold, 4), "one_at_a_time_cold_ms": round(single_cold, 4),
                     "single_factor_cold_ms": round(one_cold, 4),
                     "batch_vs_single_factor": round(batch_cold / one_cold, 3),
                     "batch_relinearize_ms": round(batch_relin, 4), "one_at_a_time_relinearize_ms": round(single_relin, 4),
                     "single_factor_relinearize_ms": round(one_relin, 4),
                     "batch_cold_mpts_s": round(nwin * per / batch_cold / 1e3, 1),
                     "without_components": {"batch_cold_ms": round(nc_batch_cold, 4), "single_factor_cold_ms": round(nc_one_cold, 4),
                                            "batch_relinearize_ms": round(nc_batch_relin, 4),
                                            "single_factor_relinearize_ms": round(nc_one_relin, 4),
                                            "batch_cold_mpts_s": round(nwin * per / nc_batch_cold / 1e3, 1),
                                            "note": "K4 skipped: H, b, f, final localizabilities and degeneracy info bit-identical; "
                                                    "component localizabilities / status histogram not produced"},
                     "note": "median wall time of synchronous raw C-ABI calls (results on the host); cold = every point of "
                             "every factor runs k-NN; relinearize = every point hits the data-association cache"}
        for f in wf:
            f.destroy()

    return {"relinearize_window": win_stats}
