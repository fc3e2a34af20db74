"""Synthetic data for the photometric path (BASELINE configs[3]): an Ouster OS0-128-style STAGGERED, skewed scan of a
textured box room, the ENWIDE photometric parameter block, and the per-column deskew poses.

Test / bench tooling only (deterministic counter-based RNG, see synth.py).  Nothing here is on the product path.

Sensor model.  Measurement (row r, column c) is stored at raw index r * cols + c (what the driver publishes) and
belongs to image column u = (c + pixel_shift_by_row[r]) % cols after destaggering (src/lidar/photometric.cpp:72-90);
image column u looks along azimuth phi(u) = pi - 2 pi (u + 0.5) / cols (descending, so that
u = fx * phi + cx with fx = -cols / 2 pi, cx = cols / 2: src/lidar/photometric_config.cpp:99-100), beams at
beam_altitude_angles[r], beam origin lidar_origin_to_beam_origin_mm off the axis; column c fires at c * col_ns.
"""
from __future__ import annotations

import numpy as np

from . import os0_128, synth

ROOM = np.array([24.0, 18.0, 6.0])
SENSOR_LOCAL = np.array([9.3, 6.6, 1.8])
T_B_L_R = synth.so3_exp(np.array([0.01, -0.02, 0.03]))
T_B_L_t = np.array([0.05, 0.02, -0.10])


def _fir(n: int, cutoff: float, highpass: bool) -> np.ndarray:
    """Windowed-sinc FIR (Hamming), unit DC gain low-pass or its spectral inversion: stands in for the shipped
    33-tap high_pass_fir / low_pass_fir tables (config/enwide/params.yaml:124-125) without copying them."""
    k = np.arange(n) - (n - 1) / 2.0
    h = np.sinc(2.0 * cutoff * k) * np.hamming(n)
    h = h / h.sum()
    if highpass:
        h = -h
        h[(n - 1) // 2] += 1.0
    return h


def photo_config(rows: int = 128, cols: int = 1024, patch: int = 5, **over) -> dict:
    """lidar.photometric block of config/enwide/params.yaml:101-135 + an OS0-128-like sensor description."""
    half = patch // 2
    if patch % 2:
        offs = [(du, dv) for dv in range(-half, half + 1) for du in range(-half, half + 1)]   # the default 5 x 5 order
    else:
        offs = [(du, dv) for dv in range(-half, half) for du in range(-half, half)]          # 8 x 8: -4 .. 3
    d = dict(
        rows=rows, cols=cols, destagger=1,
        pixel_shift_by_row=os0_128.pixel_shifts(rows, cols),                        # the sensor's own tables (config/enwide/os_enwide.json)
        beam_altitude_angles=os0_128.altitude_angles(rows).astype(np.float32),
        range_min=0.5, range_max=30.0, erosion_buffer=10, patch_size=5, margin_size=10,
        intensity_scale=0.25, intensity_gamma=1.0, remove_lines=1, filter_brightness=1, gaussian_blur=1, gaussian_blur_size=3,
        gradient_threshold=10.0, max_dist_from_mean=3.0, max_dist_from_plane=0.5, nma_radius=11, num_features_detect=60,
        occlusion_range_diff_threshold=0.2, max_feature_life_time=1000,
        high_pass_fir=_fir(33, 0.12, True), low_pass_fir=_fir(33, 0.04, False), brightness_window_size=(41, 7),
        lidar_origin_to_beam_origin_mm=os0_128.LIDAR_ORIGIN_TO_BEAM_ORIGIN_MM, rotate_patch_to_align_with_gradient=0,
        patch_offsets=np.array(offs, np.int32),
        use_robust_cost_function=0, robust_cost_function=0, robust_cost_function_parameter=1.345, error_scale=1.0,
        max_error=0.5075, sigma=0.25, T_B_L_R=T_B_L_R, T_B_L_t=T_B_L_t, static_mask=None,
    )
    d.update(over)
    return d


def _texture(hit_w: np.ndarray, face: np.ndarray) -> np.ndarray:
    """Signal photons of a return: a smooth two-scale pattern in the two in-plane coordinates of the hit face."""
    a = np.where(face == 0, hit_w[:, 1], hit_w[:, 0])
    b = np.where(face == 2, hit_w[:, 1], hit_w[:, 2])
    base = 0.5 + 0.5 * np.sin(2 * np.pi * a / 1.7) * np.sin(2 * np.pi * b / 1.1)
    fine = 0.5 + 0.5 * np.sin(2 * np.pi * (a + 0.6 * b) / 0.37)
    tiles = ((np.floor(a / 0.9) + np.floor(b / 0.9)) % 2)
    return 250.0 + 900.0 * base + 350.0 * fine * tiles


def make_frame(cfg: dict, frame: int = 0, seed: int = synth.BASE_SEED + 77, room=ROOM, v=(1.2, 0.2, 0.0), w=(0.0, 0.0, 0.35),
               col_ns: int = 97_656, dropout: float = 0.01, dt_frame: float = 0.1):
    """One staggered, skewed scan.  Returns dict(raw, deskewed, unique_ns, T_Le_Lt (n_cols x 12, fp64), R_W_L, t_W_L,
    R_W_Be, t_W_Be): raw / deskewed are POINT_DTYPE clouds with identical indexing (points_raw_ / points_full_ of
    lidar/manager.cpp:376-380, :496-509); frame k is taken dt_frame seconds of the constant twist (v, w) after frame k - 1."""
    rows, cols = cfg["rows"], cfg["cols"]
    room = np.asarray(room, float)
    v, w = np.asarray(v, float), np.asarray(w, float)
    # scan-end pose of this frame: integrate the twist from the start pose
    R_end, t_end = synth.rot_z(synth.SENSOR_YAW), synth.room_origin(0, 0) + SENSOR_LOCAL
    for _ in range(frame):
        t_end = t_end + R_end @ (v * dt_frame)
        R_end = R_end @ synth.so3_exp(w * dt_frame)
    shift = np.asarray(cfg["pixel_shift_by_row"], np.int64)
    alt = np.deg2rad(np.asarray(cfg["beam_altitude_angles"], np.float64))
    r_, c_ = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    r_, c_ = r_.ravel(), c_.ravel()
    u_ = (c_ + shift[r_]) % cols
    phi = np.pi - 2.0 * np.pi * (u_ + 0.5) / cols
    ca, sa, cz, sz = np.cos(alt[r_]), np.sin(alt[r_]), np.cos(phi), np.sin(phi)
    bo = cfg["lidar_origin_to_beam_origin_mm"] / 1000.0
    d_s = np.stack([ca * cz, ca * sz, sa], 1)
    o_s = np.stack([bo * cz, bo * sz, np.zeros_like(cz)], 1)
    t_ns = (np.arange(cols) * col_ns * (1024 // cols)).astype(np.uint32)
    t_end_ns = float(t_ns[-1])
    T = np.empty((cols, 12))
    Rc, tc = np.empty((cols, 3, 3)), np.empty((cols, 3))
    for c in range(cols):
        dt = (t_end_ns - float(t_ns[c])) * 1e-9
        R_rel, t_rel = synth.so3_exp(-w * dt), -v * dt       # T_Le_Lt: the sensor at firing time in the scan-end frame
        T[c, :9], T[c, 9:] = R_rel.ravel(), t_rel
        Rc[c], tc[c] = R_end @ R_rel, t_end + R_end @ t_rel
    d_w = np.einsum("nij,nj->ni", Rc[c_], d_s)
    o_w = tc[c_] + np.einsum("nij,nj->ni", Rc[c_], o_s)
    lo = synth.room_origin(0, 0)
    hi = lo + room
    with np.errstate(divide="ignore", invalid="ignore"):
        tpos = np.where(d_w > 0, (hi - o_w) / d_w, np.where(d_w < 0, (lo - o_w) / d_w, np.inf))
    face = tpos.argmin(1)
    rng = tpos.min(1)
    n = rows * cols
    i = np.arange(n, dtype=np.uint64) + np.uint64(frame * 1_000_003)
    rng = rng + synth._normal(seed, 7, i) * 0.01
    hit_w = o_w + rng[:, None] * d_w
    p_s = o_s + rng[:, None] * d_s
    raw = np.zeros(n, dtype=synth.POINT_DTYPE)
    raw["x"], raw["y"], raw["z"] = p_s[:, 0], p_s[:, 1], p_s[:, 2]
    raw["intensity"] = (_texture(hit_w, face) * (1.0 + 0.01 * synth._normal(seed, 9, i))).astype(np.float32)
    raw["t"] = t_ns[c_]
    raw["idx"] = (r_ * cols + c_).astype(np.uint32)
    raw["range"] = np.sqrt(raw["x"].astype(np.float32) ** 2 + raw["y"].astype(np.float32) ** 2 + raw["z"].astype(np.float32) ** 2)
    # prepareInput drops NaN returns: fewer points than pixels.  Dropouts come in 8 x 8-pixel patches (absorbing
    # surfaces), not as salt-and-pepper noise, which the 15 x 15 mask erosion would turn into an empty mask.
    tile = ((r_ // 8) * ((cols + 7) // 8) + (u_ // 8)).astype(np.uint64) + np.uint64(frame * 7_919)
    keep = synth._uniform(seed, 11, tile) >= dropout
    raw = raw[keep]
    # Manager::deskewPoints' hot loop (lidar/manager.cpp:504-508): f32 R p + t per timestamp group
    col_of = (raw["idx"] % cols).astype(np.int64)
    Rf, tf = T[:, :9].astype(np.float32).reshape(cols, 3, 3)[col_of], T[:, 9:].astype(np.float32)[col_of]
    x, y, z = raw["x"], raw["y"], raw["z"]
    desk = raw.copy()
    for k, name in enumerate(("x", "y", "z")):
        desk[name] = (Rf[:, k, 0] * x + (Rf[:, k, 1] * y + Rf[:, k, 2] * z)) + tf[:, k]
    R_W_Be = R_end @ np.asarray(cfg["T_B_L_R"]).T
    t_W_Be = t_end - R_W_Be @ np.asarray(cfg["T_B_L_t"])
    return dict(raw=raw, deskewed=desk, unique_ns=t_ns, T_Le_Lt=T, R_W_L=R_end, t_W_L=t_end, R_W_Be=R_W_Be, t_W_Be=t_W_Be)


BIAS_DIRECTIONS = np.eye(3)
