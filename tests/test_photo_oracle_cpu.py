"""CPU suite: the photometric oracle (oracle/photo_ref.hpp, C++) against the independent numpy restatement
(oracle/numpy_photo.py) on live synthetic frames, and against the committed golden fixture."""
import numpy as np

from photo_golden import check_against_golden, golden_case, rel


def test_oracle_matches_golden():
    from oracle import photo_ref

    cfg, f0, f1, g, feats = golden_case()
    check_against_golden(photo_ref.Photo(cfg), cfg, f0, f1, g, feats)


def test_oracle_vs_numpy_twin_live():
    """A different frame pair, a different pose, Huber on, a static mask, and the binary form."""
    from mimosa_amd import synth, synth_photo as sp
    from oracle import numpy_photo as npp, photo_ref

    sm = np.ones((64, 512), np.uint8)
    sm[:, 200:230] = 0                                            # e.g. the robot's own frame in view
    cfg = sp.photo_config(rows=64, cols=512, use_robust_cost_function=1, robust_cost_function_parameter=0.6, static_mask=sm.ravel())
    fa, fb = sp.make_frame(cfg, 2, seed=991), sp.make_frame(cfg, 3, seed=991)
    P = photo_ref.Photo(cfg)
    d = P.preprocess(fa["raw"], fa["deskewed"], fa["unique_ns"], fa["T_Le_Lt"])
    n = npp.preprocess(cfg, fa["raw"], fa["deskewed"], fa["unique_ns"], fa["T_Le_Lt"])
    for k in ("yaw", "range", "mask", "idx"):
        assert np.array_equal(P.image(k), n[k]), k
    assert np.array_equal(P.image("proj_idx"), n["proj_idx"])
    assert P.image("mask")[:, 193:237].sum() == 0                 # static mask + erosion
    for k in ("intensity", "dx", "dy"):
        assert np.abs(P.image(k) - n[k]).max() <= 2e-4, k
    assert np.abs(d["intensity"] - n["corrected_intensity"]).max() <= 2e-4
    P.detect(30, np.eye(3), np.zeros(3), np.eye(3))               # features in the frame's own Le frame (binary form)
    feats = P.features()
    assert len(feats) == 30
    # every stored feature is what photometric.cpp:662-706 says it is
    T_BL_R, T_BL_t = np.asarray(cfg["T_B_L_R"]), np.asarray(cfg["T_B_L_t"])
    I, idx = P.image("intensity"), P.image("idx")
    for f in feats:
        u0, v0 = int(f["center"][0]), int(f["center"][1])
        for k, (du, dv) in enumerate(cfg["patch_offsets"]):
            j = idx[v0 + dv, u0 + du]
            p = np.array([fa["deskewed"]["x"][j], fa["deskewed"]["y"][j], fa["deskewed"]["z"][j]], np.float64)
            q = T_BL_R.T @ ((T_BL_R @ p + T_BL_t) - T_BL_t)       # T_B_L^-1 * I * T_B_L * p
            assert np.abs(f["Le_ps"][k] - q).max() <= 1e-12
            assert f["intensities"][k] == I[v0 + dv, u0 + du]
        c = f["Le_ps"] - f["Le_ps"].mean(0)
        assert np.abs(c @ f["normal"]).max() <= cfg["max_dist_from_plane"] and np.dot(f["normal"], f["Le_ps"].mean(0)) < 0
        assert abs(np.linalg.norm(f["psi"]) - 1.0) <= 1e-12 and abs(f["psi"].sum()) <= 1e-12
    P.preprocess(fb["raw"], fb["deskewed"], fb["unique_ns"], fb["T_Le_Lt"])
    nb = npp.preprocess(cfg, fb["raw"], fb["deskewed"], fb["unique_ns"], fb["T_Le_Lt"])
    for k in ("intensity", "dx", "dy"):
        nb[k] = P.image(k).astype(np.float64)                     # same images: isolates the factor arithmetic
    F = P.make_factor(binary=True)
    Rb, tb = fb["R_W_Be"] @ synth.so3_exp(np.array([0.001, 0.002, -0.002])), fb["t_W_Be"] + np.array([0.01, 0.015, -0.005])
    r = F.linearize(Rb, tb, fa["R_W_Be"], fa["t_W_Be"])
    q = npp.linearize(cfg, nb, feats, Rb, tb, fa["R_W_Be"], fa["t_W_Be"])
    assert np.array_equal(r["status_hist"], q["status_hist"]) and q["status_hist"][8] >= 10
    for k in ("H_bb", "H_ba", "H_aa", "b_b", "b_a"):
        assert rel(r[k], q[k]) <= 1e-9, k
    assert abs(r["f"] - q["f"]) <= 1e-9 * q["f"]
    st, ce, rows = F.state()
    v = np.nonzero(st == 8)[0]
    assert rel(rows[v][:, :25, 0], np.array([q["e_rows"][i] for i in v])) <= 1e-9


def test_feature_circle_and_rounding_helpers():
    """The midpoint-circle fill used for non-maximum suppression: symmetric, radius-bounded, clipped at the border."""
    from mimosa_amd import synth_photo as sp
    from oracle import photo_ref

    cfg = sp.photo_config(rows=64, cols=512)
    f0 = sp.make_frame(cfg, 0)
    P = photo_ref.Photo(cfg)
    P.preprocess(f0["raw"], f0["deskewed"], f0["unique_ns"], f0["T_Le_Lt"])
    P.detect(60, f0["R_W_Be"], f0["t_W_Be"], sp.BIAS_DIRECTIONS)
    c = np.array([f["center"] for f in P.features()])
    assert len(c) > 20
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + 1e9 * np.eye(len(c))
    assert d.min() > cfg["nma_radius"] - 1
    m = P.image("detection_mask")
    assert np.all(m[c[:, 1].astype(int), c[:, 0].astype(int)] == 1)   # detections lie inside the eroded mask
    g = P.image("grad")
    assert np.all(g[c[:, 1].astype(int), c[:, 0].astype(int)] > cfg["gradient_threshold"])
