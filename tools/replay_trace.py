#!/usr/bin/env python3
"""Run the sequence replay (for rocprofv3 --kernel-trace --memory-copy-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, replay
ctx = capi.Context(0)
cfg = replay.ReplayConfig(n_scans=20)
scans = replay.make_scans(cfg)
r = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
print("scans/s", r["scans_per_s"], r["stage_s"])
