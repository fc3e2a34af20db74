"""-m gpu: the photometric path (BASELINE configs[3]) — Photometric::preprocess, detectFeatures,
updateMap and PhotometricFactor::linearize through the C ABI (mh_photo_*) against the CPU oracle (oracle/photo_ref.hpp)
on identical synthetic frames: a 128 x 1024 staggered, skewed OS0-128-style scan of a textured room."""
import os
import sys

import numpy as np
import pytest

from parity import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    from mimosa_amd import synth_photo as sp

    cfg = sp.photo_config()
    return cfg, [sp.make_frame(cfg, k) for k in range(3)]


def _both(ctx, cfg):
    from mimosa_amd import capi
    from oracle import photo_ref

    return capi.Photo(ctx, cfg), photo_ref.Photo(cfg)


def _pre(P, f):
    return P.preprocess(f["raw"], f["deskewed"], f["unique_ns"], f["T_Le_Lt"])


def _assert_images(g, r):
    # integer / byte images and the f32 filter chain: bit for bit
    for name in ("range", "mask", "idx", "intensity", "dx", "dy", "grad", "detection_mask"):
        a, b = g.image(name), r.image(name)
        assert np.array_equal(a, b), (name, int((a != b).sum()))
    # yaw: atan2 evaluated in fp64 by two different libms, then rounded to f32: identical up to a handful of 1-ulp cases
    ya, yb = g.image("yaw"), r.image("yaw")
    bad = ya != yb
    assert bad.sum() <= 8 and np.abs(ya - yb).max() <= 5e-7, int(bad.sum())
    # proj_idx depends on round(project(p)): identical unless a projection sits on a rounding tie
    pa, pb = g.image("proj_idx"), r.image("proj_idx")
    assert (np.any(pa != pb, axis=2)).sum() <= 4


def test_preprocess_matches_oracle(ctx, frames):
    cfg, fr = frames
    g, r = _both(ctx, cfg)
    for f in fr[:2]:
        dg, dr = _pre(g, f), _pre(r, f)
        _assert_images(g, r)
        assert dg.tobytes() == dr.tobytes()                     # corrected intensities written back into the cloud
        assert not np.array_equal(dg["intensity"], f["deskewed"]["intensity"])
    pj = g.image("proj_idx")
    assert (pj[:, :, 0] > 1).sum() > 500 and pj[:, :, 0].max() <= 9      # duplicates exist and are capped
    assert 0.5 < g.image("mask").mean() < 1.0
    g.destroy()


def _same_features(fa, fb, tol=1e-9):
    assert len(fa) == len(fb)
    for a, b in zip(fa, fb):
        assert a["id"] == b["id"] and a["life_time"] == b["life_time"]
        assert np.array_equal(a["center"], b["center"])
        assert np.abs(a["Le_ps"] - b["Le_ps"]).max() <= tol
        assert np.array_equal(a["intensities"], b["intensities"])
        assert np.abs(a["psi"] - b["psi"]).max() <= tol
        assert min(np.abs(a["normal"] - b["normal"]).max(), np.abs(a["normal"] + b["normal"]).max()) <= 1e-7
        assert np.dot(a["normal"], b["normal"]) > 0.999                  # same "towards the sensor" orientation


def test_detect_features_matches_oracle(ctx, frames):
    from mimosa_amd import synth_photo as sp

    cfg, fr = frames
    g, r = _both(ctx, cfg)
    _pre(g, fr[0]), _pre(r, fr[0])
    for P in (g, r):
        P.detect(25, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    assert len(g.features()) == 25
    _same_features(g.features(), r.features())
    for P in (g, r):                                                     # a second call keeps clear of the tracked ones
        P.detect(35, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS[:2])
    fa = g.features()
    assert len(fa) == 60
    _same_features(fa, r.features())
    c = np.array([f["center"] for f in fa])
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + 1e9 * np.eye(len(c))
    assert d.min() > cfg["nma_radius"] - 1                               # non-maximum suppression radius respected
    g.destroy()


def _assert_factor(gr, rr, gs, rs, binary=False):
    assert np.array_equal(gr["status_hist"], rr["status_hist"]), (gr["status_hist"], rr["status_hist"])
    assert gr["n_exceptions"] == rr["n_exceptions"]
    assert np.array_equal(gs[0], rs[0])
    v = gs[0] == 8
    assert v.sum() >= 10
    assert np.abs(gs[1][v] - rs[1][v]).max() <= 1e-8                      # new centres (sub-pixel)
    tol = 1e-5  # north_star: within 1e-5 relative
    assert rel(gr["H_bb"], rr["H_bb"]) <= tol and rel(gr["b_b"], rr["b_b"]) <= tol
    assert abs(gr["f"] - rr["f"]) <= tol * abs(rr["f"])
    if binary:
        assert rel(gr["H_ba"], rr["H_ba"]) <= tol and rel(gr["H_aa"], rr["H_aa"]) <= tol and rel(gr["b_a"], rr["b_a"]) <= tol
    else:
        assert rel(gr["loc_rot_final"], rr["loc_rot_final"]) <= tol and rel(gr["loc_trans_final"], rr["loc_trans_final"]) <= tol
    # per patch point: whitened NCC residual and Jacobian row (north_star "8x8 patch residuals + image Jacobians")
    ga, ra = gs[2][v], rs[2][v]
    assert np.array_equal(ga[:, :, 7], ra[:, :, 7])
    assert rel(ga[:, :, 0], ra[:, :, 0]) <= tol
    assert rel(ga[:, :, 1:7], ra[:, :, 1:7]) <= tol


def _tracked(ctx, cfg, fr, n_detect=60):
    from mimosa_amd import synth_photo as sp

    g, r = _both(ctx, cfg)
    _pre(g, fr[0]), _pre(r, fr[0])
    for P in (g, r):
        P.detect(n_detect, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    _pre(g, fr[1]), _pre(r, fr[1])
    return g, r


def test_unary_factor_parity(ctx, frames):
    from mimosa_amd import synth

    cfg, fr = frames
    g, r = _tracked(ctx, cfg, fr)
    gf, rf = g.make_factor(), r.make_factor()
    poses = [(fr[1]["R_W_Be"], fr[1]["t_W_Be"]),
             (fr[1]["R_W_Be"] @ synth.so3_exp(np.array([0.002, -0.001, 0.003])), fr[1]["t_W_Be"] + np.array([0.02, -0.01, 0.01])),
             (fr[1]["R_W_Be"] @ synth.so3_exp(np.array([-0.004, 0.003, -0.002])), fr[1]["t_W_Be"] + np.array([-0.03, 0.02, 0.0]))]
    for R, t in poses:
        gr, rr = gf.linearize(R, t), rf.linearize(R, t)
        assert rr["status_hist"][8] >= 30
        _assert_factor(gr, rr, gf.state(), rf.state())
    # V S V^T of Photometric::getFactors: only the degenerate directions of the geometric factor are constrained
    V = synth.so3_exp(np.array([0.3, -0.2, 0.5]))
    V6 = np.block([[V, np.zeros((3, 3))], [np.zeros((3, 3)), V.T]])
    S = np.diag([1.0, 0, 1, 0, 1, 1])
    VSVt = V6 @ S @ V6.T
    gf2, rf2 = g.make_factor(VSVt), r.make_factor(VSVt)
    R, t = poses[1]
    gr, rr = gf2.linearize(R, t), rf2.linearize(R, t)
    assert np.array_equal(gr["status_hist"], rr["status_hist"])
    assert rel(gr["H_bb"], rr["H_bb"]) <= 1e-5 and rel(gr["b_b"], rr["b_b"]) <= 1e-5
    # the enqueue / collect split (what the smoother uses next to mh_icp_linearize_batch) == the blocking call
    gf2.linearize_async(R, t)
    with pytest.raises(Exception, match="in flight"):
        gf2.linearize_async(R, t)
    ga = gf2.wait()
    assert np.array_equal(ga["H_bb"], gr["H_bb"]) and np.array_equal(ga["b_b"], gr["b_b"]) and ga["f"] == gr["f"]
    assert np.array_equal(ga["status_hist"], gr["status_hist"])
    # clone() (ISAM2 clones factors): an independent factor on the same frame, same bits; survives its source
    gc = gf2.clone()
    assert np.array_equal(gc.state(rows=False)[0], gf2.state(rows=False)[0])
    gf2.destroy()
    g._pre_frame = None
    gr2 = gc.linearize(R, t)
    assert np.array_equal(gr2["H_bb"], gr["H_bb"]) and np.array_equal(gr2["b_b"], gr["b_b"]) and gr2["f"] == gr["f"]
    for f in (gf, gc):
        f.destroy()
    g.destroy()


def test_binary_factor_parity(ctx, frames):
    from mimosa_amd import synth

    cfg, fr = frames
    g, r = _both(ctx, cfg)
    _pre(g, fr[0]), _pre(r, fr[0])
    I3, z3 = np.eye(3), np.zeros(3)
    for P in (g, r):                      # binary form: features live in the Le frame of frame a (T_W_Be = identity at detection)
        P.detect(40, I3, z3, np.eye(3))
    _pre(g, fr[1]), _pre(r, fr[1])
    gf, rf = g.make_factor(binary=True), r.make_factor(binary=True)
    Ra, ta = fr[0]["R_W_Be"], fr[0]["t_W_Be"]
    Rb, tb = fr[1]["R_W_Be"] @ synth.so3_exp(np.array([0.001, 0.002, -0.002])), fr[1]["t_W_Be"] + np.array([0.01, 0.015, -0.005])
    gr, rr = gf.linearize(Rb, tb, Ra, ta), rf.linearize(Rb, tb, Ra, ta)
    _assert_factor(gr, rr, gf.state(), rf.state(), binary=True)
    gf.destroy()
    g.destroy()


def test_8x8_patches_configs3(ctx):
    """BASELINE configs[3] words the patches as 8 x 8 (the reference default is 5 x 5): 64 points per
    feature, one wave lane each."""
    from mimosa_amd import synth, synth_photo as sp

    cfg = sp.photo_config(patch=8)
    fr = [sp.make_frame(cfg, k) for k in range(2)]
    g, r = _tracked(ctx, cfg, fr)
    assert all(len(f["Le_ps"]) == 64 for f in g.features())
    gf, rf = g.make_factor(), r.make_factor()
    R, t = fr[1]["R_W_Be"] @ synth.so3_exp(np.array([0.002, 0.001, -0.001])), fr[1]["t_W_Be"] + np.array([0.01, 0.02, -0.01])
    _assert_factor(gf.linearize(R, t), rf.linearize(R, t), gf.state(), rf.state())
    gf.destroy()
    g.destroy()


def test_update_map_bookkeeping(ctx, frames):
    """Photometric::updateMap over three frames: invalid features dropped, centres / life times updated, the missing
    ones re-detected — the tracked set stays identical to the oracle's."""
    from mimosa_amd import synth_photo as sp

    cfg, fr = frames
    g, r = _both(ctx, cfg)
    for k, f in enumerate(fr):
        _pre(g, f), _pre(r, f)
        gf = rf = None
        if g.features():
            gf, rf = g.make_factor(), r.make_factor()
            gr, rr = gf.linearize(f["R_W_Be"], f["t_W_Be"]), rf.linearize(f["R_W_Be"], f["t_W_Be"])
            assert np.array_equal(gr["status_hist"], rr["status_hist"])
        g.update_map(gf, f["R_W_Be"], f["t_W_Be"], sp.BIAS_DIRECTIONS)
        r.update_map(rf, f["R_W_Be"], f["t_W_Be"], sp.BIAS_DIRECTIONS)
        fa, fb = g.features(), r.features()
        assert len(fa) == len(fb) and len(fa) > 40
        for a, b in zip(fa, fb):
            assert a["id"] == b["id"] and a["life_time"] == b["life_time"]
            assert np.abs(a["center"] - b["center"]).max() <= 1e-8
        if gf is not None:
            gf.destroy()
        if k == 2:
            assert max(f_["life_time"] for f_ in fa) == 3
    g.destroy()


def test_ragged_inputs(ctx, frames):
    """Empty cloud, everything out of range, an image row without a single return, too many points."""
    from mimosa_amd import capi

    cfg, fr = frames
    g, r = _both(ctx, cfg)
    f = fr[0]
    e = f["raw"][:0]
    for P in (g, r):
        P.preprocess(e, e, f["unique_ns"], f["T_Le_Lt"])
    _assert_images(g, r)
    assert g.image("mask").sum() == 0
    far = f["deskewed"].copy()
    far["range"] = 99.0                                         # beyond range_max: no image content, yaw table still built
    for P in (g, r):
        P.preprocess(f["raw"], far, f["unique_ns"], f["T_Le_Lt"])
    _assert_images(g, r)
    keep = (f["raw"]["idx"] // cfg["cols"]) != 17                 # row 17 never returns: linear yaw ramp +pi .. -pi
    for P in (g, r):
        P.preprocess(f["raw"][keep], f["deskewed"][keep], f["unique_ns"], f["T_Le_Lt"])
    _assert_images(g, r)
    with pytest.raises(capi.MhError):
        big = np.concatenate([f["raw"], f["raw"][:4000]])
        g.preprocess(big, big, f["unique_ns"], f["T_Le_Lt"])
    with pytest.raises(capi.MhError):
        capi.Photo(ctx, dict(cfg, gaussian_blur_size=5))        # a deliberate limit
    g.destroy()


@pytest.mark.parametrize("erosion_buffer", [10, 0])
def test_rotated_patches(ctx, frames, erosion_buffer):
    """rotate_patch_to_align_with_gradient (photometric.cpp:659-684, photometric_utils.cpp:453-518): every new feature samples
    the pattern rotated into its edge frame and snapped to distinct pixels.  With erosion_buffer 0 a rotated pattern can reach
    pixels the eroded mask does not vouch for: those candidates are skipped on both sides (the reference's behaviour is
    undefined there)."""
    from mimosa_amd import capi, synth, synth_photo as sp

    cfg0, fr = frames
    cfg = dict(cfg0, rotate_patch_to_align_with_gradient=1, erosion_buffer=erosion_buffer)
    g, r = _both(ctx, cfg)
    _pre(g, fr[0]), _pre(r, fr[0])
    for P in (g, r):
        P.detect(40, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    fa = g.features()
    assert len(fa) >= 20
    _same_features(fa, r.features())
    plain, _ = _both(ctx, dict(cfg, rotate_patch_to_align_with_gradient=0))
    _pre(plain, fr[0])
    plain.detect(40, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    fp = plain.features()
    n = min(len(fa), len(fp))
    assert any(not np.array_equal(a["intensities"], b["intensities"]) for a, b in zip(fa[:n], fp[:n]))    # the patterns really moved
    _pre(g, fr[1]), _pre(r, fr[1])
    gf, rf = g.make_factor(), r.make_factor()
    R, t = fr[1]["R_W_Be"] @ synth.so3_exp(np.array([0.002, -0.001, 0.003])), fr[1]["t_W_Be"] + np.array([0.02, -0.01, 0.01])
    _assert_factor(gf.linearize(R, t), rf.linearize(R, t), gf.state(), rf.state())
    gf.destroy()
    g.destroy()
    plain.destroy()


def test_preprocess_from_resident_scan(ctx, frames):
    """mh_photo_preprocess_scan: raw cloud and deskewed cloud taken from a device-resident mh_scan."""
    from mimosa_amd import capi, synth

    cfg, fr = frames
    f = fr[0]
    raw = np.zeros(len(f["raw"]), dtype=synth.OUSTER_DTYPE)
    for k in ("x", "y", "z", "intensity", "t"):
        raw[k] = f["raw"][k]
    # prepareInput wants the full raw grid in index order to reproduce idx = position: rebuild it with NaN holes
    full = np.zeros(cfg["rows"] * cfg["cols"], dtype=synth.OUSTER_DTYPE)
    full["x"] = np.nan
    full[f["raw"]["idx"]] = raw
    full["ring"] = (np.arange(len(full)) // cfg["cols"]).astype(np.uint16)
    sc = capi.Scan(ctx)
    ctx.check(ctx.L.mh_scan_keep_raw(sc.h, 1))
    info = sc.prepare_input(full, capi.make_input_config(range_min=0.0, range_max=1000.0))
    assert info["n_full"] == len(f["raw"])
    uns = sc.unique_ns()
    sel = np.searchsorted(f["unique_ns"], uns)
    sc.deskew(f["T_Le_Lt"][sel].astype(np.float32))
    g, g2 = capi.Photo(ctx, cfg), capi.Photo(ctx, cfg)
    got_full = sc.points(0)                                      # deskewed on the device, intensities still raw
    g2.preprocess_scan(sc, f["T_Le_Lt"][sel])
    corrected = g.preprocess(f["raw"], got_full, uns, f["T_Le_Lt"][sel])  # the host-buffer entry point on the same two clouds
    assert sc.points(0).tobytes() == corrected.tobytes()         # the resident cloud received the corrected intensities
    for name in ("range", "mask", "idx", "intensity", "dx", "dy", "yaw", "proj_idx"):
        assert np.array_equal(g.image(name), g2.image(name)), name
    g.destroy()
    g2.destroy()
    sc.destroy()


def _resident_scan(ctx, cfg, f):
    from mimosa_amd import capi, synth

    raw = np.zeros(len(f["raw"]), dtype=synth.OUSTER_DTYPE)
    for k in ("x", "y", "z", "intensity", "t"):
        raw[k] = f["raw"][k]
    full = np.zeros(cfg["rows"] * cfg["cols"], dtype=synth.OUSTER_DTYPE)
    full["x"] = np.nan
    full[f["raw"]["idx"]] = raw
    full["ring"] = (np.arange(len(full)) // cfg["cols"]).astype(np.uint16)
    sc = capi.Scan(ctx)
    ctx.check(ctx.L.mh_scan_keep_raw(sc.h, 1))
    sc.prepare_input(full, capi.make_input_config(range_min=0.0, range_max=1000.0))
    sel = np.searchsorted(f["unique_ns"], sc.unique_ns())
    sc.deskew(f["T_Le_Lt"][sel].astype(np.float32))
    return sc, f["T_Le_Lt"][sel]


def test_preprocess_begin_commit_beside_update_map(ctx, frames):
    """mh_photo_preprocess_scan_begin / _commit: the frame of scan k is built (on another host thread's time) while
    mh_photo_update_map of scan k - 1 runs on the same object, and becomes current afterwards — features, images and the
    factor must be the bits of the plain sequence preprocess(k-1), update_map(k-1), preprocess(k)."""
    import threading
    from mimosa_amd import capi
    from mimosa_amd import synth_photo as sp

    cfg, fr = frames
    sc0, T0 = _resident_scan(ctx, cfg, fr[0])
    sc1, T1 = _resident_scan(ctx, cfg, fr[1])
    plain, piped = capi.Photo(ctx, cfg), capi.Photo(ctx, cfg)
    plain.preprocess_scan(sc0, T0)
    plain.detect(60, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    plain.preprocess_scan(sc1, T1)
    want = plain.features()
    with pytest.raises(capi.MhError):
        piped.preprocess_commit()                       # nothing begun
    for rep in range(3):                                # the overlap is a matter of timing: a few rounds
        sc0b, _ = _resident_scan(ctx, cfg, fr[0])       # (fresh scans: preprocess writes the corrected intensities into them)
        sc1b, _ = _resident_scan(ctx, cfg, fr[1])
        piped.set_features([])
        piped.preprocess_scan(sc0b, T0)
        if rep:                                         # the frame-only part of the detection enqueued ahead of it
            piped.detect_prefetch()
            if rep == 2:
                piped.detect_prefetch()                 # twice is once
        errors = []

        def update():
            try:
                piped.detect(60, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        th = threading.Thread(target=update)
        th.start()
        piped.preprocess_scan_begin(sc1b, T1)          # beside the detection on frame 0
        th.join()
        assert not errors, errors
        piped.preprocess_commit()
        got = piped.features()
        assert len(got) == len(want) > 0
        for a, b in zip(got, want):
            for k in a:
                if k != "id":                                        # ids keep counting across the rounds
                    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        for name in ("range", "mask", "idx", "intensity", "dx", "dy", "yaw", "proj_idx"):
            assert np.array_equal(plain.image(name), piped.image(name)), name
        assert sc1.points(0).tobytes() == sc1b.points(0).tobytes()
        sc0b.destroy()
        sc1b.destroy()
    for o in (plain, piped, sc0, sc1):
        o.destroy()


def test_preprocess_begin_twice_and_left_pending(ctx, frames):
    """A begun frame that is never committed is dropped by the next begin (the commit then installs the LATER one) and by
    destroy; the scan's cloud is only written at the commit."""
    from mimosa_amd import capi

    cfg, fr = frames
    sc0, T0 = _resident_scan(ctx, cfg, fr[0])
    sc1, T1 = _resident_scan(ctx, cfg, fr[1])
    ref1 = capi.Photo(ctx, cfg)
    sc1r, _ = _resident_scan(ctx, cfg, fr[1])
    ref1.preprocess_scan(sc1r, T1)
    p = capi.Photo(ctx, cfg)
    before = sc1.points(0).tobytes()
    p.preprocess_scan_begin(sc0, T0)
    p.preprocess_scan_begin(sc1, T1)                   # drops the frame of scan 0
    assert sc1.points(0).tobytes() == before            # not written yet
    p.preprocess_commit()
    assert sc1.points(0).tobytes() == sc1r.points(0).tobytes()
    for name in ("range", "mask", "idx", "intensity", "dx", "dy", "yaw", "proj_idx"):
        assert np.array_equal(p.image(name), ref1.image(name)), name
    with pytest.raises(capi.MhError):
        p.preprocess_commit()                           # consumed
    p.preprocess_scan_begin(sc0, T0)                    # left pending: destroy must cope
    for o in (p, ref1, sc0, sc1, sc1r):
        o.destroy()


def test_hip_matches_golden_fixture(ctx):
    """The committed fixture (generated by the independent numpy restatement) pins the HIP path as it pins the oracle."""
    from mimosa_amd import capi
    from photo_golden import check_against_golden, golden_case

    cfg, f0, f1, g, feats = golden_case()
    P = capi.Photo(ctx, cfg)
    F = check_against_golden(P, cfg, f0, f1, g, feats)
    F.destroy()
    P.destroy()


def test_single_stage_kernels_still_match_the_oracle():
    """Round 4 runs the image chain, the mask erosion and the yaw-table fill as ONE launch (photo_chain_kernel) where the
    configuration fits its strip scheme; the single-stage kernels stay as the general path.  MH_PHOTO_UNFUSED=1 forces them:
    the preprocess / golden / resident-scan tests of this file pass either way (a fresh process: the switch is read once)."""
    import subprocess
    env = dict(os.environ, MH_PHOTO_UNFUSED="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                          "preprocess_matches_oracle or golden_fixture or resident_scan or ragged"], capture_output=True, text=True, env=env, timeout=900,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
