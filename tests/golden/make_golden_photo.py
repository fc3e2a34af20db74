#!/usr/bin/env python3
"""Generates tests/golden/photo_64x512.npz: expected outputs of the photometric path on a deterministic synthetic
frame pair (mimosa_amd.synth_photo, 64 x 512 image), computed by the INDEPENDENT numpy restatement
(oracle/numpy_photo.py).  The feature set comes from the C++ oracle's detectFeatures (its std::sort tie order has no
numpy counterpart) and is stored as data; the factor outputs for those features are numpy's.  Run from the repo root:
    python tests/golden/make_golden_photo.py
The inputs are not stored (they are regenerated from the counter-based RNG); their SHA-256 is, to detect drift."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mimosa_amd import synth, synth_photo as sp  # noqa: E402
from oracle import numpy_photo as npp, photo_ref  # noqa: E402


def case():
    cfg = sp.photo_config(rows=64, cols=512)
    f0, f1 = sp.make_frame(cfg, 0), sp.make_frame(cfg, 1)
    R = f1["R_W_Be"] @ synth.so3_exp(np.array([0.002, -0.001, 0.003]))
    t = f1["t_W_Be"] + np.array([0.02, -0.01, 0.01])
    return cfg, f0, f1, R, t


def main():
    cfg, f0, f1, R, t = case()
    n0 = npp.preprocess(cfg, f0["raw"], f0["deskewed"], f0["unique_ns"], f0["T_Le_Lt"])
    n1 = npp.preprocess(cfg, f1["raw"], f1["deskewed"], f1["unique_ns"], f1["T_Le_Lt"])
    P = photo_ref.Photo(cfg)
    P.preprocess(f0["raw"], f0["deskewed"], f0["unique_ns"], f0["T_Le_Lt"])
    P.detect(40, f0["R_W_Be"], f0["t_W_Be"], sp.BIAS_DIRECTIONS)
    feats = P.features()
    q = npp.linearize(cfg, n1, feats, R, t)
    valid = q["status"] == 8
    out = dict(
        inputs_sha256=np.frombuffer(hashlib.sha256(f0["raw"].tobytes() + f0["deskewed"].tobytes() + f1["raw"].tobytes()).digest(), np.uint8),
        intensity0=n0["intensity"].astype(np.float32), mask0=np.packbits(n0["mask"]), yaw0_rows=n0["yaw"][[0, 17, 63]],
        yaw0_sum=np.float64(n0["yaw"].astype(np.float64).sum()), proj_count0=n0["proj_idx"][:, :, 0].astype(np.uint8),
        idx0_sum=np.int64(n0["idx"].astype(np.int64).sum()), dx0_row=n0["dx"][31].astype(np.float32), dy0_row=n0["dy"][31].astype(np.float32),
        feat_id=np.array([f["id"] for f in feats]), feat_center=np.array([f["center"] for f in feats]),
        feat_Le_ps=np.array([f["Le_ps"] for f in feats]), feat_psi=np.array([f["psi"] for f in feats]),
        feat_intensities=np.array([f["intensities"] for f in feats]), feat_normal=np.array([f["normal"] for f in feats]),
        lin_R=R, lin_t=t, status=q["status"], centers=q["centers"], H_bb=q["H_bb"], b_b=q["b_b"], f=np.float64(q["f"]),
        loc_rot_final=q["loc_rot_final"], loc_trans_final=q["loc_trans_final"],
        e_rows=np.array([q["e_rows"][i] for i in np.nonzero(valid)[0]]), J_rows=np.array([q["J_rows"][i] for i in np.nonzero(valid)[0]]),
    )
    path = os.path.join(ROOT, "tests", "golden", "photo_64x512.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; valid features:", int(valid.sum()), "of", len(feats))


if __name__ == "__main__":
    main()
