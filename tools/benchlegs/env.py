"""Shared objects of one bench.py run, handed to the side legs (tools/benchlegs/*.py)."""
import os
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
INFLIGHT = 32
INFLIGHT_ICP = 64  # calls of one unsharded factor in flight in the headline loop (the library's limit per factor, kMaxPending)


class Env(SimpleNamespace):
    """args, rank / local_rank / world / dist, ctx / ctxs, gmap, factor / factors, pts, R, t, cfgd, n_pts, room_clouds, capi, synth,
    barrier(), run_steps(k, collect, fs), raw_linearize(t), the pre-marshalled _R / _g / _out of the raw C-ABI call, _all_reduce,
    and `results`: what the legs that already ran have returned (the sharded leg reads relinearize_window from there)."""


def c_sync_ns(ctx, factor_h, R, t, g, out, n, reset=True):
    """Wall time (ns, one per call) of n synchronous mh_icp_linearize calls made FROM C (tools/micro/sync_caller.c: clock_gettime
    around the foreign call alone; before each call the association state is reset when `reset` and the stream is drained) —
    what the reference's C++ caller would pay.  A ctypes call from Python adds 1-2 us of argument conversion."""
    import ctypes as C

    import numpy as np

    from mimosa_amd import build as hb
    H = C.CDLL(hb.build_sync_caller())
    vp = C.c_void_p
    H.mh_sync_caller_run.argtypes = [vp] * 9 + [C.c_int, C.c_int, vp]
    fn = lambda f: C.cast(f, vp)
    L = ctx.L
    Rm, tv, gv = (np.ascontiguousarray(x, np.float64) for x in (R, t, g))
    ns = np.zeros(n, np.float64)
    rc = H.mh_sync_caller_run(fn(L.mh_icp_linearize), fn(L.mh_icp_reset), fn(L.mh_synchronize), ctx.h, factor_h, Rm.ctypes.data_as(vp),
                              tv.ctypes.data_as(vp), gv.ctypes.data_as(vp), C.byref(out), n, 1 if reset else 0, ns.ctypes.data_as(vp))
    assert rc == 0, rc
    return ns


def make_sync_stepper(ctx, factor_h, R, t, g):
    """A closure run(outs): len(outs) cold synchronous mh_icp_linearize calls made FROM C back to back (tools/micro/sync_caller.c:
    mh_sync_caller_steps: reset + linearize per step, result i in outs[i]); `outs` = a ctypes array of capi.IcpResult.
    Everything that is not the calls — loading the harness, marshalling the arguments — happens here, once."""
    import ctypes as C

    import numpy as np

    from mimosa_amd import build as hb
    H = C.CDLL(hb.build_sync_caller())
    vp = C.c_void_p
    H.mh_sync_caller_steps.argtypes = [vp] * 7 + [C.c_size_t, C.c_int]
    L = ctx.L
    lin, rst = C.cast(L.mh_icp_linearize, vp), C.cast(L.mh_icp_reset, vp)
    Rm, tv, gv = (np.ascontiguousarray(x, np.float64) for x in (R, t, g))
    pR, pt, pg = Rm.ctypes.data_as(vp), tv.ctypes.data_as(vp), gv.ctypes.data_as(vp)
    fn = H.mh_sync_caller_steps

    def run(outs, _keep=(Rm, tv, gv)):
        n = len(outs)
        rc = fn(lin, rst, factor_h, pR, pt, pg, C.cast(outs, vp), C.sizeof(outs) // max(n, 1), n)
        assert rc == 0, rc
    return run
