#!/usr/bin/env python3
"""Wall time of mh_scan_prefetch (pinned staging copy + async upload) alone and of the calls around it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mimosa_amd import capi, synth
ctx = capi.Context(0)
raw, aux = synth.make_raw_scan(128)
cfg = capi.make_input_config()
a, b = capi.Scan(ctx), capi.Scan(ctx)
for s in (a, b):
    s.prefetch(raw); s.prepare_input_prefetched(cfg)
def t(fn, n=30):
    v = []
    for _ in range(n):
        ctx.synchronize(); x = time.perf_counter(); fn(); v.append(time.perf_counter() - x)
    return round(float(np.median(v)) * 1e6, 1)
print("prefetch_us", t(lambda: a.prefetch(raw)))
print("prefetch+prepare_us", t(lambda: (a.prefetch(raw), a.prepare_input_prefetched(cfg))))
print("prepare_input(host buffer)_us", t(lambda: b.prepare_input(raw, cfg)))
import threading
# prefetch on a thread while the main thread runs prepare on the other handle
def both():
    th = threading.Thread(target=lambda: a.prefetch(raw)); th.start(); b.prepare_input(raw, cfg); th.join()
print("threaded prefetch || prepare_us", t(both))
tmp = np.empty_like(raw)
print("host memcpy 4MiB_us", t(lambda: np.copyto(tmp, raw)))
