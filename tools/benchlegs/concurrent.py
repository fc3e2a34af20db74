"""bench.py side leg (tools/benchlegs): several scans in flight on separate contexts (value_concurrent)

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
    # ---- several scans in flight: 4 contexts (one HIP stream each) sharing the map, in a process of their own ----
    # HIP multiplexes a process's streams onto 4 hardware queues and two busy streams that share one serialise
    # (profiles/r04_concurrency_bisect.md); this process already holds contexts of its own, so four MORE here would share
    # queues.  tools/conc_probe.py runs the same loop — cold linearizes dealt round-robin to the streams by one host thread,
    # <= 32 in flight per stream — in a fresh process with one stream per context.
    conc = None
    if args.streams == 1 and args.concurrent_streams > 1 and not args.profile_mode and world == 1:
        import subprocess
        env = dict(os.environ)
        here = ROOT
        try:
            pr = subprocess.run([sys.executable, os.path.join(here, "tools", "conc_probe.py"), "--streams", str(args.concurrent_streams),
                                 "--steps", str(max(400, args.steps * 2))], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")][-1]
            pj = json.loads(line)
            conc = {"streams": pj["streams"], "steps": pj["steps"], "value": pj["conc_mpts"], "ms_per_step": pj["conc_ms"],
                    "single_stream_same_process_ms": pj["single_ms"], "one_host_thread_per_stream_mpts": pj["threads_mpts"],
                    "note": "tools/conc_probe.py in a process of its own (best of 3 passes of this many steps): 4 contexts, one HIP stream each, "
                            "sharing one map"}
        except Exception as e:  # the figure is a side leg: say why it is missing
            conc = {"error": f"{type(e).__name__}: {e}"}

    return {"value_concurrent": conc}
