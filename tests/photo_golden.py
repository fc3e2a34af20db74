"""Shared checks of a photometric implementation (C++ oracle or the HIP path) against tests/golden/photo_64x512.npz."""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photo_64x512.npz")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def golden_case():
    from mimosa_amd import synth, synth_photo as sp

    cfg = sp.photo_config(rows=64, cols=512)
    f0, f1 = sp.make_frame(cfg, 0), sp.make_frame(cfg, 1)
    g = np.load(GOLD)
    sha = np.frombuffer(hashlib.sha256(f0["raw"].tobytes() + f0["deskewed"].tobytes() + f1["raw"].tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, g["inputs_sha256"]), "synthetic inputs drifted: regenerate the fixture"
    feats = [dict(id=int(g["feat_id"][i]), life_time=1, center=g["feat_center"][i], normal=g["feat_normal"][i], mean_intensity=0.0,
                  sigma_intensity=1.0, Le_ps=g["feat_Le_ps"][i], intensities=g["feat_intensities"][i], psi=g["feat_psi"][i])
             for i in range(len(g["feat_id"]))]
    return cfg, f0, f1, g, feats


def check_against_golden(P, cfg, f0, f1, g, feats, exact_images=False):
    """P: a Photo-like object (preprocess / image / set_features / make_factor)."""
    P.preprocess(f0["raw"], f0["deskewed"], f0["unique_ns"], f0["T_Le_Lt"])
    I = P.image("intensity")
    assert np.abs(I.astype(np.float64) - g["intensity0"]).max() <= 1e-3 and rel(I, g["intensity0"]) <= 1e-6   # f32 chain vs f64 twin
    assert np.array_equal(np.packbits(P.image("mask")), g["mask0"])
    yaw = P.image("yaw")
    assert np.abs(yaw[[0, 17, 63]] - g["yaw0_rows"]).max() <= 5e-7 and abs(yaw.astype(np.float64).sum() - g["yaw0_sum"]) <= 1e-3
    assert np.array_equal(P.image("proj_idx")[:, :, 0].astype(np.uint8), g["proj_count0"])
    assert int(P.image("idx").astype(np.int64).sum()) == int(g["idx0_sum"])
    assert np.abs(P.image("dx")[31] - g["dx0_row"]).max() <= 1e-3 and np.abs(P.image("dy")[31] - g["dy0_row"]).max() <= 1e-3
    P.preprocess(f1["raw"], f1["deskewed"], f1["unique_ns"], f1["T_Le_Lt"])
    P.set_features(feats)
    F = P.make_factor()
    r = F.linearize(g["lin_R"], g["lin_t"])
    st, ce, rows = F.state()
    assert np.array_equal(st, g["status"])
    v = st == 8
    assert np.abs(ce[v] - g["centers"][v]).max() <= 1e-6
    # the factor reads the f32 image of this implementation, the fixture used the f64-filtered twin image: 1e-5 relative
    assert rel(r["H_bb"], g["H_bb"]) <= 1e-5 and rel(r["b_b"], g["b_b"]) <= 1e-5 and abs(r["f"] - g["f"]) <= 1e-5 * g["f"]
    assert rel(r["loc_rot_final"], g["loc_rot_final"]) <= 1e-5 and rel(r["loc_trans_final"], g["loc_trans_final"]) <= 1e-5
    m = g["e_rows"].shape[1]
    assert rel(rows[v][:, :m, 0], g["e_rows"]) <= 1e-5
    assert rel(rows[v][:, :m, 1:7], g["J_rows"]) <= 1e-5
    return F
