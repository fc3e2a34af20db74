#!/usr/bin/env python3
"""Soak of the pipelined native replay: N runs of the 20-scan sequence, each compared bit for bit (poses, costs, tracked
features per scan) with the sequential loop's result.  Prints one JSON line."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import replay
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = replay.ReplayConfig(n_scans=20, rows=128)
scans = replay.make_scans(cfg)
with tempfile.TemporaryDirectory() as td:
    ref = replay.run_native(cfg, scans, td, repeats=1, sequential=True)
    def key(r):
        poses = b"".join(np.asarray(R, np.float64).tobytes() + np.asarray(t, np.float64).tobytes() for R, t in r["poses_est"])
        return poses, json.dumps(r.get("photo_valid")), np.asarray(r["first_costs"], np.float64).tobytes()
    want, bad, rates = key(ref), 0, []
    for i in range(n):
        r = replay.run_native(cfg, scans, td, repeats=2)
        rates.append(r["scans_per_s"])
        bad += key(r) != want
print(json.dumps({"runs": n, "different_from_sequential": bad, "scans_per_s_min": round(min(rates), 1), "scans_per_s_median": round(float(np.median(rates)), 1),
                  "scans_per_s_max": round(max(rates), 1)}))
