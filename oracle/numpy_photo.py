"""ORACLE — TEST INFRASTRUCTURE ONLY.  Independent numpy / scipy restatement of the photometric path, written from
the reference sources without looking at oracle/photo_ref.hpp's loops: it pins the C++ oracle (CPU suite) and
generates the golden fixtures (tests/golden/make_golden_photo.py).  PARITY UNPINNED — see photo_ref.hpp for the
OpenCV / Eigen / PCL assumptions (O1-O12); this file makes the same assumptions with different machinery:
scipy.ndimage with mode="mirror" (= BORDER_REFLECT_101) evaluated in float64, np.linalg for the small algebra,
explicit J_psi matrices.  Agreement with the C++ oracle is therefore to ~1e-6 relative on the f32 image chain (float32
vs float64 accumulation) and ~1e-12 on the fp64 factor given the same images.

Reference: src/lidar/photometric.cpp:92-371, include/mimosa/lidar/photometric_factor.hpp:136-355,
src/lidar/photometric_utils.cpp:13-388 (paths relative to /root/reference/mimosa/).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

DEG2RAD, RAD2DEG = 0.017453293, 57.29578  # PCL's macros (pcl/pcl_macros.h); the reference defines neither
DUP = 10


def derived(cfg):
    """src/lidar/photometric_config.cpp:98-110"""
    cols, rows = cfg["cols"], cfg["rows"]
    alt = np.asarray(cfg["beam_altitude_angles"], np.float32)
    fx = -float(np.float32(cols)) / (2 * np.pi)
    cx = float(np.float32(cols)) / 2.0
    fy = -float(np.float32(rows)) / abs(float(np.float32(alt[0] - alt[-1])) * DEG2RAD)
    bo = float(np.float32(cfg["lidar_origin_to_beam_origin_mm"] / 1000.0))
    return fx, fy, cx, bo


def pixel_of(cfg, idx):
    """photometric.cpp:72-90 inverted: raw index -> (u, v)"""
    cols = cfg["cols"]
    v = idx // cols
    c = idx % cols
    shift = np.asarray(cfg["pixel_shift_by_row"], np.int64)
    u = np.where(cfg["destagger"], (c + shift[v]) % cols, c)
    return u.astype(np.int64), v.astype(np.int64)


def yaw_table(cfg, raw):
    """photometric.cpp:121-199"""
    rows, cols = cfg["rows"], cfg["cols"]
    yaw = np.full((rows, cols), np.nan, np.float32)
    u, v = pixel_of(cfg, raw["idx"].astype(np.int64))
    yaw[v, u] = np.arctan2(raw["y"].astype(np.float64), raw["x"].astype(np.float64)).astype(np.float32)
    f32 = np.float32
    for r in range(rows):
        ok = np.nonzero(~np.isnan(yaw[r]))[0]
        if len(ok) == 0:
            t = np.arange(cols, dtype=f32) / f32(cols - 1)
            yaw[r] = ((f32(1) - t).astype(np.float64) * np.pi + t.astype(np.float64) * (-np.pi)).astype(f32)
            continue
        first, last = ok[0], ok[-1]
        if first > 0:
            t = np.arange(first, dtype=f32) / f32(first)
            yaw[r, :first] = ((f32(1) - t).astype(np.float64) * np.pi + (t * yaw[r, first]).astype(np.float64)).astype(f32)
        for a, b in zip(ok[:-1], ok[1:]):
            if b - a > 1:
                t = np.arange(1, b - a, dtype=f32) / f32(b - a)
                yaw[r, a + 1:b] = yaw[r, a] + t * (yaw[r, b] - yaw[r, a])
        if last < cols - 1:
            gap = cols - 1 - last
            t = np.arange(1, gap + 1, dtype=f32) / f32(gap)
            yaw[r, last + 1:] = (((f32(1) - t) * yaw[r, last]).astype(np.float64) + t.astype(np.float64) * (-np.pi)).astype(f32)
    return yaw


def project(cfg, p, yaw):
    """photometric_utils.cpp:80-198, one point.  Returns (ok, u, v)."""
    fx, fy, cx, bo = derived(cfg)
    alt = np.asarray(cfg["beam_altitude_angles"], np.float32)
    rows, cols = cfg["rows"], cfg["cols"]
    L = np.sqrt(p[0] * p[0] + p[1] * p[1]) - bo
    R = np.sqrt(L * L + p[2] * p[2])
    phi, theta = np.arctan2(p[1], p[0]), np.arcsin(p[2] / R)
    u = fx * phi + cx
    if u < 0 or u >= cols:
        raise ValueError("Invalid x coordinate")
    if u < 5 or u > cols - 5:
        return False, u, 0.0
    if theta > float(alt[0]) * DEG2RAD or theta < float(alt[-1]) * DEG2RAD:
        return False, u, 0.0
    th = theta * RAD2DEG
    above = np.nonzero(alt.astype(np.float64) > th)[0]  # last table entry above the angle; clamped where the reference
    g = min(int(above[-1]) if len(above) else 0, rows - 2)  # would step outside the table (DEG2RAD * RAD2DEG != 1)
    v = g + (float(alt[g]) - th) / float(np.float32(alt[g] - alt[g + 1]))
    row = yaw[int(np.round(v))]
    il, ir = int(u) - 5, int(u) + 5
    while ir - il > 1:
        mid = il + (ir - il) // 2
        if float(row[mid]) == phi:
            return (0 <= mid <= cols - 1 and 0 <= v <= rows - 1), float(mid), v
        if float(row[mid]) < phi:
            ir = mid
        else:
            il = mid
    u = il + (float(row[il]) - phi) / float(np.float32(row[il] - row[ir]))
    return (0 <= u <= cols - 1 and 0 <= v <= rows - 1), u, v


def round_half_away(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def preprocess(cfg, raw, desk, unique_ns, T_Le_Lt):
    """Photometric::preprocess (photometric.cpp:92-320).  Returns a dict of images + the corrected intensities."""
    rows, cols = cfg["rows"], cfg["cols"]
    yaw = yaw_table(cfg, raw)
    inr = (desk["range"] >= np.float32(cfg["range_min"])) & (desk["range"] <= np.float32(cfg["range_max"]))
    u, v = pixel_of(cfg, desk["idx"].astype(np.int64))
    I = np.zeros((rows, cols), np.float32)
    rng = np.zeros((rows, cols), np.float32)
    mask = np.zeros((rows, cols), np.uint8)
    idx = np.full((rows, cols), -1, np.int32)
    sel = np.nonzero(inr)[0]
    I[v[sel], u[sel]] = desk["intensity"][sel]
    rng[v[sel], u[sel]] = desk["range"][sel]
    mask[v[sel], u[sel]] = 1
    idx[v[sel], u[sel]] = sel
    proj = np.zeros((rows, cols, DUP), np.int32)
    for i in sel:  # photometric.cpp:218-244
        ok, pu, pv = project(cfg, np.array([desk["x"][i], desk["y"][i], desk["z"][i]], np.float64), yaw)
        if not ok:
            continue
        uu, vv = int(round_half_away(pu)), int(round_half_away(pv))
        if proj[vv, uu, 0] + 1 < DUP:
            proj[vv, uu, 0] += 1
            proj[vv, uu, proj[vv, uu, 0]] = i
    img = I.astype(np.float64)
    if cfg["intensity_scale"] != 1.0:
        img = (I * np.float32(cfg["intensity_scale"])).astype(np.float64)
    if cfg["intensity_gamma"] != 1.0:
        img = img ** float(cfg["intensity_gamma"])
    if cfg["remove_lines"]:  # :322-337, correlation (ndimage.correlate1d does not flip), REFLECT_101 = "mirror"
        hp = ndimage.correlate1d(img, np.asarray(cfg["high_pass_fir"], np.float32).astype(np.float64), axis=0, mode="mirror")
        lines = ndimage.correlate1d(hp, np.asarray(cfg["low_pass_fir"], np.float32).astype(np.float64), axis=1, mode="mirror")
        img = np.maximum(img - lines, 0.0)
    if cfg["filter_brightness"]:  # :339-347; cv::Size(width, height)
        w, h = cfg["brightness_window_size"]
        b = ndimage.uniform_filter(img, size=(h, w), mode="mirror") + 1.0
        img = 140.0 * img / b
    if cfg["gaussian_blur"]:  # ksize 3, sigma 0 -> {0.25, 0.5, 0.25}
        k = np.array([0.25, 0.5, 0.25])
        img = ndimage.correlate1d(ndimage.correlate1d(img, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    img = np.minimum(img, 255.0)
    corrected = desk["intensity"].astype(np.float64).copy()
    has = idx >= 0
    corrected[idx[has]] = img[has]
    dx = ndimage.correlate1d(img, np.array([-0.5, 0.0, 0.5]), axis=1, mode="mirror")
    dy = ndimage.correlate1d(img, np.array([-0.5, 0.0, 0.5]), axis=0, mode="mirror")
    if cfg.get("static_mask") is not None:
        mask = mask & (np.asarray(cfg["static_mask"]).reshape(rows, cols) != 0)
    k = cfg["patch_size"] + cfg["erosion_buffer"]
    # anchor k // 2: offsets -k//2 .. k-1-k//2, which is scipy's window for both parities; out-of-image = 1 (never lowers)
    em = ndimage.minimum_filter(mask, size=(k, k), mode="constant", cval=1)
    return dict(yaw=yaw, intensity=img, range=rng, mask=em.astype(np.uint8), idx=idx, proj_idx=proj, dx=dx, dy=dy,
                corrected_intensity=corrected, points=desk, pose_ns=np.asarray(unique_ns), pose_T=np.asarray(T_Le_Lt, float).reshape(-1, 12))


def bilinear(img, x, y):
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    dx, dy = x - x0, y - y0
    return ((1 - dx) * (1 - dy) * img[y0, x0] + dx * (1 - dy) * img[y0, x0 + 1] + (1 - dx) * dy * img[y0 + 1, x0] +
            dx * dy * img[y0 + 1, x0 + 1])


def projection_jacobian(cfg, p):
    fx, fy, cx, bo = derived(cfg)
    rxy = np.hypot(p[0], p[1])
    L = rxy - bo
    R2 = L * L + p[2] * p[2]
    return np.array([[-fx * p[1] / rxy**2, fx * p[0] / rxy**2, 0.0],
                     [-fy * p[0] * p[2] / ((L + bo) * R2), -fy * p[1] * p[2] / ((L + bo) * R2), fy * L / R2]])


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def pose(R, t):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def project_undistorted(cfg, fr, p):
    """photometric_utils.cpp:287-366.  Returns (ok, Li_p, (u, v), R_Le_Lt)."""
    ok, ku, kv = project(cfg, p, fr["yaw"])
    if not ok:
        return False, None, None, None
    col, row = int(round_half_away(ku)), int(round_half_away(kv))
    pj = fr["proj_idx"]
    if pj[row, col, 0] == 0:
        rr = np.nonzero(pj[:, col, 0] > 0)[0]
        if len(rr) == 0:
            return False, None, None, None
        row = int(rr[0])
    n = pj[row, col, 0]
    cand = pj[row, col, 1:1 + n]
    pts = fr["points"]
    if n > 1:
        d = np.array([np.float32(np.sum((p - np.array([pts["x"][j], pts["y"][j], pts["z"][j]], np.float64)) ** 2)) for j in cand])
        j = int(cand[int(np.argmin(d))])  # first minimum, like the strict "<" of the reference
    else:
        j = int(cand[0])
    k = np.searchsorted(fr["pose_ns"], pts["t"][j])
    if k >= len(fr["pose_ns"]) or fr["pose_ns"][k] != pts["t"][j]:
        raise KeyError("interpolated_map_T_Le_Lt.at")
    R, t = fr["pose_T"][k, :9].reshape(3, 3), fr["pose_T"][k, 9:]
    Li = R.T @ (p - t)
    ok, u, v = project(cfg, Li, fr["yaw"])
    return ok, Li, (u, v), R


def linearize(cfg, fr, features, R_b, t_b, R_a=None, t_a=None, VSVt=None):
    """PhotometricFactor::linearize (photometric_factor.hpp:136-355) with explicit matrices."""
    binary = R_a is not None
    T_BL = pose(np.asarray(cfg["T_B_L_R"], float), np.asarray(cfg["T_B_L_t"], float))
    Tb = pose(R_b, t_b)
    Ta = pose(R_a, t_a) if binary else np.eye(4)
    dBe = np.linalg.inv(Tb) @ Ta
    dLe = np.linalg.inv(T_BL) @ dBe @ T_BL
    rows, cols, margin = cfg["rows"], cfg["cols"], cfg["margin_size"]
    Hbb, Hba, Haa, bb, ba, fsum = np.zeros((6, 6)), np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(6), np.zeros(6), 0.0
    statuses, centers, e_rows, J_rows = [], [], [], []
    for ft in features:
        m = len(ft["Le_ps"])
        st = 0
        uv, pl, Rl, Ib = [], [], [], []
        for i in range(m):
            p = dLe[:3, :3] @ ft["Le_ps"][i] + dLe[:3, 3]
            ok, Li, q, R = project_undistorted(cfg, fr, p)
            if not ok:
                st = 1
                break
            r = np.linalg.norm(Li)
            if r < cfg["range_min"] or r > cfg["range_max"]:
                st = 2
                break
            ux, uy = int(round_half_away(q[0])), int(round_half_away(q[1]))
            if not fr["mask"][uy, ux]:
                st = 4
                break
            if ux < margin or ux >= cols - margin or uy < margin or uy >= rows - margin:
                st = 5
                break
            if abs(float(fr["range"][uy, ux]) - r) > float(np.float32(cfg["occlusion_range_diff_threshold"])):
                st = 6
                break
            uv.append(q), pl.append(Li), Rl.append(R), Ib.append(bilinear(fr["intensity"], *q))
        if st:
            statuses.append(st), centers.append(ft["center"]), e_rows.append(None), J_rows.append(None)
            continue
        Ib = np.array(Ib)
        mean = Ib.mean()
        sigma = np.linalg.norm(Ib - mean)
        psi = (Ib - mean) / sigma
        e = psi - ft["psi"]
        if (2 - e @ e) / 2 < cfg["max_error"]:
            statuses.append(7), centers.append(ft["center"]), e_rows.append(None), J_rows.append(None)
            continue
        statuses.append(8), centers.append(np.array(uv[m // 2]))
        Db, Da = np.zeros((m, 6)), np.zeros((m, 6))
        for i in range(m):
            g = np.array([bilinear(fr["dx"], *uv[i]), bilinear(fr["dy"], *uv[i])]) @ projection_jacobian(cfg, pl[i])
            p_a = T_BL[:3, :3] @ ft["Le_ps"][i] + T_BL[:3, 3]
            p_b = dBe[:3, :3] @ p_a + dBe[:3, 3]
            Rk = Rl[i].T @ T_BL[:3, :3].T
            Db[i] = g @ np.hstack([Rk @ hat(p_b), -Rk])
            if binary:
                Rka = Rk @ dBe[:3, :3]
                Da[i] = g @ np.hstack([-Rka @ hat(p_a), Rka])
        Jpsi = ((np.eye(m) - np.outer(psi, psi)) / sigma) @ (np.eye(m) - np.ones((m, m)) / m)
        Jb, Ja = Jpsi @ Db, Jpsi @ Da
        wh = np.linalg.norm(e) / cfg["sigma"]
        sw = 1.0
        if cfg["use_robust_cost_function"]:
            c = cfg["robust_cost_function_parameter"]
            sw = (1.0 if abs(wh) <= c else np.sqrt(c / abs(wh))) if cfg["robust_cost_function"] == 0 else c * c / (c * c + wh * wh)
        Jb, Ja, e = Jb * sw / cfg["sigma"], Ja * sw / cfg["sigma"], e * sw / cfg["sigma"]
        Hbb += Jb.T @ Jb
        bb += Jb.T @ e
        fsum += e @ e
        if binary:
            Haa += Ja.T @ Ja
            Hba += Jb.T @ Ja
            ba += Ja.T @ e
        e_rows.append(e), J_rows.append(Jb)
    out = dict(status=np.array(statuses, np.int32), centers=np.array(centers, float), f=fsum, e_rows=e_rows, J_rows=J_rows,
               status_hist=np.bincount(statuses, minlength=9))
    if binary:
        out.update(H_bb=Hbb, H_ba=Hba, H_aa=Haa, b_b=bb, b_a=ba)
        return out
    V = np.eye(6) if VSVt is None else np.asarray(VSVt, float)
    H = V @ Hbb @ V
    out.update(H_bb=H, b_b=H @ np.linalg.inv(Hbb) @ bb)
    wr, Er = np.linalg.eigh(H[:3, :3])
    wt, Et = np.linalg.eigh(H[3:, 3:])
    with np.errstate(invalid="ignore"):
        out.update(loc_rot_final=np.sqrt(wr), loc_trans_final=np.sqrt(wt), eigvec_rot=Er, eigvec_trans=Et)
    return out


def gradient_based_locations(grad_x, grad_y, pattern):
    """getGradientBasedLocations + snapPoint (src/lidar/photometric_utils.cpp:453-518), written independently of
    oracle/photo_ref.hpp: the pattern rotated into the (edge normal, edge tangent) frame, every point snapped to the nearest
    free pixel (round half away from zero; when taken, the nearest of the 8 neighbours in dx-major, dy-minor order)."""
    gx, gy = np.float32(grad_x), np.float32(grad_y)
    norm = float(np.sqrt(np.float32(gx * gx + gy * gy))) + 1e-6          # float products, float sqrt, then double
    nx, ny, tx, ty = -float(gy) / norm, float(gx) / norm, float(gx) / norm, float(gy) / norm
    used, out = set(), []
    for px, py in pattern:
        rx, ry = nx * float(px) + tx * float(py), ny * float(px) + ty * float(py)
        g = (int(round_half_away(rx)), int(round_half_away(ry)))
        if g in used:
            best, best_d = g, float("inf")
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if dx == 0 and dy == 0:
                        continue
                    alt = (g[0] + dx, g[1] + dy)
                    if alt in used:
                        continue
                    d = float(np.sqrt((alt[0] - rx) ** 2 + (alt[1] - ry) ** 2))
                    if d < best_d:
                        best, best_d = alt, d
            g = best
        used.add(g)
        out.append(g)
    return np.array(out, np.int32)
