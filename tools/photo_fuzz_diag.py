#!/usr/bin/env python3
"""Diagnose one seed of tests/test_gpu_photo_fuzz.py: which patch rows differ between HIP and the oracle."""
import os, sys
import numpy as np
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from mimosa_amd import capi, synth, synth_photo as sp
from oracle import photo_ref
import test_gpu_photo_fuzz as tf
from test_gpu_photo import _pre
np.set_printoptions(linewidth=200, precision=12)
seed = int(sys.argv[1])
ctx = capi.Context(0)
cfg, fr, rng = tf._case(seed)
print({k: cfg[k] for k in ("rows", "cols", "patch_size", "use_robust_cost_function", "robust_cost_function", "max_error", "sigma", "error_scale", "occlusion_range_diff_threshold")}, "n_off", len(cfg["patch_offsets"]))
g, r = capi.Photo(ctx, cfg), photo_ref.Photo(cfg)
_pre(g, fr[0]); _pre(r, fr[0])
n_det = int(rng.choice([10, 40]))
dirs = sp.BIAS_DIRECTIONS[: int(rng.integers(1, len(sp.BIAS_DIRECTIONS) + 1))]
binary = bool(rng.integers(0, 2))
Rd, td = (np.eye(3), np.zeros(3)) if binary else (fr[0]["R_W_Be"], fr[0]["t_W_Be"])
for P in (g, r):
    P.detect(n_det, Rd, td, dirs)
_pre(g, fr[1]); _pre(r, fr[1])
gf, rf = g.make_factor(binary=binary), r.make_factor(binary=binary)
for it in range(2):
    R = fr[1]["R_W_Be"] @ synth.so3_exp(rng.normal(0, 0.003, 3))
    t = fr[1]["t_W_Be"] + rng.normal(0, 0.02, 3)
    args = (R, t, fr[0]["R_W_Be"], fr[0]["t_W_Be"]) if binary else (R, t)
    gr, rr = gf.linearize(*args), rf.linearize(*args)
    gs, rs = gf.state(), rf.state()
    print("it", it, "binary", binary, "hist", list(gr["status_hist"]), list(rr["status_hist"]))
    for key in ("b_b", "f", "b_a"):
        print("   ", key, "hip", np.asarray(gr[key]).ravel()[:6], "ref", np.asarray(rr[key]).ravel()[:6])
    print("    H_bb diag hip", np.diag(np.asarray(gr["H_bb"]).reshape(6, 6)), "ref", np.diag(np.asarray(rr["H_bb"]).reshape(6, 6)))
    for f_ in np.nonzero(gs[0] != rs[0])[0]:
        print(" STATUS differs: feature", f_, gs[0][f_], rs[0][f_], "\n  hip res", gs[2][f_, :12, 0], "\n  ref res", rs[2][f_, :12, 0], "\n  hip J0", gs[2][f_, :3, 1:7], "\n  ref J0", rs[2][f_, :3, 1:7])
    d = np.abs(gs[2] - rs[2]).max(axis=2)
    bad = np.argwhere(d > 1e-9)
    for f_, p_ in bad[:8]:
        print(" feature", f_, "point", p_, "status", gs[0][f_], rs[0][f_], "centre", gs[1][f_], rs[1][f_])
        print("  hip", gs[2][f_, p_])
        print("  ref", rs[2][f_, p_])
    if len(bad):
        f_ = bad[0][0]
        print(" all residuals of feature: hip", gs[2][f_, :, 0], "\n ref", rs[2][f_, :, 0], "\n flags", gs[2][f_, :, 7], rs[2][f_, :, 7])
