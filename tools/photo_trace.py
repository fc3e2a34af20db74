#!/usr/bin/env python3
"""The photometric workload for rocprofv3 --kernel-trace --stats: preprocess of a 128 x 1024 frame (host buffers),
feature detection, 30 factor linearizations (60 features, 8 x 8 patches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, synth, synth_photo as sp

ctx = capi.Context(0)
cfg = sp.photo_config(patch=int(os.environ.get("MH_PATCH", "8")))
fr = [sp.make_frame(cfg, k) for k in range(2)]
G = capi.Photo(ctx, cfg)
for rep in range(10):
    G.preprocess(fr[0]["raw"], fr[0]["deskewed"], fr[0]["unique_ns"], fr[0]["T_Le_Lt"])
G.detect(60, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
G.preprocess(fr[1]["raw"], fr[1]["deskewed"], fr[1]["unique_ns"], fr[1]["T_Le_Lt"])
F = G.make_factor()
R, t = fr[1]["R_W_Be"] @ synth.so3_exp(np.array([0.002, -0.001, 0.003])), fr[1]["t_W_Be"] + np.array([0.02, -0.01, 0.01])
for rep in range(30):
    r = F.linearize(R, t)
print("done", r["status_hist"])
