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
