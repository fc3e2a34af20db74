#!/usr/bin/env python3
"""Diagnose one seed of tests/test_gpu_fuzz.py::test_random_configuration: which points' cached plane differs between the
HIP factor and the oracle, and what their neighbourhood looks like (eigenvalues of the k-point covariance)."""
import os, sys
import numpy as np
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from mimosa_amd import capi, synth
from oracle import ref_cpu
import test_gpu_fuzz as tf

seed = int(sys.argv[1])
ctx = capi.Context(0)
m, pts, cfg, mode, R, t, rng = tf._case(seed)
print({k: cfg[k] for k in ("reg_4_dof", "project_on_degneneracy", "degen_thresh_rot", "degen_thresh_trans", "use_huber", "target_ivox_map_leaf_size", "target_ivox_map_min_dist_in_voxel", "num_corres_points", "max_corres_distance", "plane_validity_distance")}, "mode", mode, "n", len(pts))
leaf, md = cfg["target_ivox_map_leaf_size"], cfg["target_ivox_map_min_dist_in_voxel"]
gm = capi.VoxelMap(ctx, leaf=leaf, min_dist=md, mode=mode)
rm = ref_cpu.Map(leaf=leaf, min_dist=md, mode=mode)
for chunk in np.array_split(m, 2):
    gm.insert(chunk); rm.insert(chunk)
gf = capi.ICPFactor(ctx, gm, pts, capi.make_reg_config(**cfg))
rf = ref_cpu.ICP(rm, pts, ref_cpu.make_config(**cfg))
g_unit = np.array([0.02, -0.01, -1.0]); g_unit /= np.linalg.norm(g_unit)
thr = md / 4.0
k = cfg["num_corres_points"]
for step in range(4):
    got, ref = gf.linearize(R, t, g_unit), rf.linearize(R, t, g_unit)
    for key in ("H_ss", "b_s", "loc_trans_final", "loc_rot_final", "loc_trans_comp", "loc_rot_comp", "degen_rot", "degen_trans"):
        a, b = np.asarray(got[key], float), np.asarray(ref[key], float)
        r = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
        if not (r <= 1e-9):
            print("  ", key, "rel", r, "\n    hip", a.ravel()[:9], "\n    ref", b.ravel()[:9])
    if step == 0 or True:
        H = np.asarray(ref["H_ss"]).reshape(6, 6)
        print("   eig(H)", np.linalg.eigvalsh(H), "hist", list(ref["status_hist"]), list(got["status_hist"]), "f", got["f"], ref["f"])
    sg, sr = gf.state(), rf.state()
    dn = np.abs(sg[2] - sr[2]).max(axis=1)
    dm = np.abs(sg[1] - sr[1]).max(axis=1)
    bad = np.nonzero((dn > 1e-9) | (dm > 1e-9) | (sg[0] != sr[0]))[0]
    print("step", step, "bad", len(bad), "H rel", np.linalg.norm(np.asarray(got["H_ss"]) - np.asarray(ref["H_ss"])) / np.linalg.norm(ref["H_ss"]))
    xyz = synth.points_xyz(pts).astype(np.float64)
    for i in bad[:6]:
        q = R @ xyz[i] + t
        nb, sq, found = gm.knn(q[None], k)
        P = nb[0][: found[0]]
        c = np.cov(P.T) if len(P) > 1 else np.zeros((3, 3))
        w = np.linalg.eigvalsh(c)
        print(" i", i, "status", sg[0][i], sr[0][i], "\n  n_hip", sg[2][i], "\n  n_ref", sr[2][i], "\n  dot", float(sg[2][i] @ sr[2][i]), "dmean", dm[i], "\n  eig", w, "ratio w1/w0", w[1] / max(w[0], 1e-300), "\n  sqd", sq[0])
    t = t + rng.normal(0, 1.0, 3) * thr * rng.choice([0.3, 1.0, 3.0])
    R = R @ synth.so3_exp(rng.normal(0, 1.0, 3) * thr / 5.0)
