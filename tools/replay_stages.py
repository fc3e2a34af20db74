#!/usr/bin/env python3
"""Main-thread stage split of replay_native (pipelined and sequential) on the bench's 20-scan sequence, ms per scan."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import replay
cfg = replay.ReplayConfig(n_scans=20, rows=128)
scans = replay.make_scans(cfg)
out = {}
for name, kw in (("pipelined", {}), ("sequential", dict(sequential=True))):
    with tempfile.TemporaryDirectory() as td:
        r = replay.run_native(cfg, scans, td, repeats=3, **kw)
    out[name] = {"scans_per_s": round(r["scans_per_s"], 1), "ms_per_scan": round(1e3 / r["scans_per_s"], 3),
                 "stages": {k: round(v / cfg.n_scans * 1e3, 3) for k, v in r["stage_s"].items()},
                 **{k: round(v / cfg.n_scans * 1e3, 3) for k, v in r["detail_s"].items()}}
print(json.dumps(out, indent=1))
