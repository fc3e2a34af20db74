"""Sequence replay harness (SURVEY.md §8 "next" row f-4, BASELINE configs[4]): deskew + geometric + photometric factors
per scan in a fixed-lag window, end to end through the C ABI, on a synthetic trajectory (ENWIDE bags and ROS are absent).

Every scan goes through the reference's LiDAR call order (src/lidar/manager.cpp:45-147):

    prepareInput -> [IMU propagation over the distinct timestamps] -> deskewPoints -> Photometric::preprocess ->
    Geometric::preprocess -> getFactors (ICPFactor ctor, PhotometricFactor ctor) -> define: smoother update +
    additional_update_iterations, EVERY live ICPFactor re-linearized each time (src/graph/manager.cpp:585-588) ->
    Geometric::updateMap (keyframe test on max |yaw, pitch, roll|, copy-then-insert) + Photometric::updateMap

What stands in for the parts that are out of scope (GTSAM / ISAM2, the IMU manager):
  * IMU: gyro / specific-force samples of the true constant-twist motion (+ noise) at `imu_rate`; propagated on the host
    exactly as Manager::deskewPoints does between samples (constant acc / omega extrapolation, manager.cpp:478-489) to get
    the per-timestamp deskew poses T_Le_Lt AND the relative-pose measurement between consecutive scans;
  * smoother: a dense Gauss-Newton over the `window` most recent poses: one unary ICP Hessian factor per live scan (all of
    them re-linearized per iteration through ONE mh_icp_linearize_batch call), the photometric factor on the newest pose,
    between-factors from the IMU propagation, a prior on the oldest pose (what marginalisation leaves behind).  Retraction
    T <- T Exp(xi), xi = (omega, v) in the body frame: the perturbation the reference's Jacobians are taken against
    (geometric_factor.hpp:341-355, photometric_factor.hpp:262-279).

`backend` does the numerical work: HipBackend drives the C ABI; tests/ run the same loop on the CPU oracle
(oracle/replay_backend.py) and require the same trajectory.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import synth, synth_photo

GRAVITY = np.array([0.0, 0.0, -9.81])


def _photo_cfg(rows, cols):
    return synth_photo.photo_config(rows=rows, cols=cols, T_B_L_R=np.eye(3), T_B_L_t=np.zeros(3))


@dataclass
class ReplayConfig:
    n_scans: int = 20
    rows: int = 128
    cols: int = 1024
    dt: float = 0.1                                   # 10 Hz
    v: tuple = (1.2, 0.2, 0.0)                        # body-frame twist of the platform
    w: tuple = (0.0, 0.0, 0.35)
    room: tuple = tuple(synth_photo.ROOM)
    window: int = 5                                   # smoother lag 0.5 s at 10 Hz (config/enwide/params.yaml:30)
    update_iters: int = 6                             # smoother_->update + additional_update_iterations: 5 (:52)
    imu_rate: float = 200.0
    imu_gyro_noise: float = 2e-4                      # rad/s, per sample
    imu_acc_noise: float = 5e-3                       # m/s^2
    prior_trans_noise: float = 0.03                   # error of the very first pose guess, metres
    prior_rot_noise_deg: float = 0.3
    between_sigma_rot: float = 2e-3                   # weight of the IMU relative-pose factor
    between_sigma_trans: float = 1e-2
    keyframe_trans_thresh: float = 1.0                # ENWIDE: map_keyframe_trans_thresh
    keyframe_rot_thresh_deg: float = 20.0
    photometric: bool = True
    reg: dict = field(default_factory=synth.enwide_config)
    photo: dict = None

    def __post_init__(self):
        if self.photo is None:
            self.photo = _photo_cfg(self.rows, self.cols)


# ---- SE(3) helpers (Pose3 tangent order: rotation first) ---------------------------------------------------------
def _hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _so3_log(R):
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return w * (0.5 if th < 1e-9 else th / (2.0 * np.sin(th)))


def _retract(R, t, xi):
    """T <- T * Exp(xi), xi = (omega, v), first order in the translation (as in round 1)."""
    return R @ synth.so3_exp(np.asarray(xi[:3])), t + R @ np.asarray(xi[3:])


def _between(Ra, ta, Rb, tb):
    return Ra.T @ Rb, Ra.T @ (tb - ta)


def _adjoint(R, t):
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[3:, 3:] = R
    A[3:, :3] = _hat(t) @ R
    return A


def trajectory(cfg: ReplayConfig):
    """Ground-truth scan-END sensor poses (what synth_photo.make_frame integrates)."""
    R, t = synth.rot_z(synth.SENSOR_YAW), synth.room_origin(0, 0) + synth_photo.SENSOR_LOCAL
    v, w = np.asarray(cfg.v, float), np.asarray(cfg.w, float)
    out = []
    for _ in range(cfg.n_scans):
        out.append((R.copy(), t.copy()))
        t = t + R @ (v * cfg.dt)
        R = R @ synth.so3_exp(w * cfg.dt)
    return out


def imu_samples(cfg: ReplayConfig, R_end, k, seed=5):
    """Gyro / accelerometer samples over the sweep of scan k: times tau_k - dt + j h, j = 0..M (scan-end time tau_k = k dt).
    Constant body twist: omega = w, specific force f = w x v - R(t)^T g."""
    M = int(round(cfg.imu_rate * cfg.dt))
    h = cfg.dt / M
    v, w = np.asarray(cfg.v, float), np.asarray(cfg.w, float)
    ts = k * cfg.dt - cfg.dt + h * np.arange(M + 1)
    rng = np.random.default_rng(seed * 7919 + k)
    gyro = np.tile(w, (M + 1, 1)) + cfg.imu_gyro_noise * rng.standard_normal((M + 1, 3))
    acc = np.empty((M + 1, 3))
    for j, tj in enumerate(ts):
        Rj = R_end @ synth.so3_exp(w * (tj - k * cfg.dt))
        acc[j] = np.cross(w, v) - Rj.T @ GRAVITY
    acc += cfg.imu_acc_noise * rng.standard_normal((M + 1, 3))
    return ts, gyro, acc


def make_scans(cfg: ReplayConfig):
    """Per scan: the raw Ouster grid the driver publishes (NaN where there was no return), the frame of
    synth_photo.make_frame (ground truth, exact deskew poses — used by tests only), and the IMU samples of the sweep."""
    scans = []
    pc = cfg.photo
    for k in range(cfg.n_scans):
        fr = synth_photo.make_frame(pc, frame=k, v=cfg.v, w=cfg.w, room=np.asarray(cfg.room), dt_frame=cfg.dt)
        grid = np.zeros(cfg.rows * cfg.cols, dtype=synth.OUSTER_DTYPE)
        grid["x"] = np.nan
        sub = np.zeros(len(fr["raw"]), dtype=synth.OUSTER_DTYPE)
        for name in ("x", "y", "z", "intensity", "t"):
            sub[name] = fr["raw"][name]
        grid[fr["raw"]["idx"]] = sub
        grid["ring"] = (np.arange(len(grid)) // cfg.cols).astype(np.uint16)
        grid["t"] = np.tile(fr["unique_ns"], cfg.rows)
        scans.append(dict(raw=grid, frame=fr, imu=imu_samples(cfg, fr["R_W_L"], k), R_gt=fr["R_W_L"], t_gt=fr["t_W_L"],
                          header_ts=k * cfg.dt - float(fr["unique_ns"][-1]) * 1e-9))
    return scans


def propagate(R, p, vel, imu, header_ts, unique_ns):
    """Manager::deskewPoints' host part (src/lidar/manager.cpp:455-499): states at the IMU sample times by integrating
    sample to sample, then constant-acc / omega extrapolation to every distinct timestamp.  (R, p, vel) = state at the
    first sample.  Returns (T_W_Bt per timestamp as (n, 12), state at the last sample)."""
    ts, gyro, acc = imu
    states = [(R, p, vel)]
    for c in range(len(ts) - 1):
        Rc, pc, vc = states[-1]
        d = ts[c + 1] - ts[c]
        aw = Rc @ acc[c] + GRAVITY
        states.append((Rc @ synth.so3_exp(gyro[c] * d), pc + vc * d + 0.5 * aw * d * d, vc + aw * d))
    # every distinct timestamp: sample interval c with ts[c] < tq <= ts[c + 1] (manager.cpp:470-476), vectorised
    tq = header_ts + np.asarray(unique_ns, np.float64) * 1e-9
    ci = np.clip(np.searchsorted(ts[1:-1], tq, side="left"), 0, len(ts) - 2)
    Rs = np.stack([s_[0] for s_ in states])[ci]
    ps = np.stack([s_[1] for s_ in states])[ci]
    vs = np.stack([s_[2] for s_ in states])[ci]
    d = (tq - ts[ci])[:, None]
    wv = gyro[ci] * d
    th = np.linalg.norm(wv, axis=1)
    th2 = th * th
    with np.errstate(invalid="ignore", divide="ignore"):
        A = np.where(th < 1e-12, 1.0 - th2 / 6.0, np.sin(th) / th)
        B = np.where(th < 1e-12, 0.5 - th2 / 24.0, (1.0 - np.cos(th)) / th2)
    K = np.zeros((len(tq), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -wv[:, 2], wv[:, 1], wv[:, 2], -wv[:, 0], -wv[:, 1], wv[:, 0]
    E = np.eye(3)[None] + A[:, None, None] * K + B[:, None, None] * (K @ K)
    out = np.empty((len(tq), 12))
    out[:, :9] = (Rs @ E).reshape(-1, 9)
    out[:, 9:] = ps + vs * d + 0.5 * np.einsum("nij,nj->ni", Rs, acc[ci]) * d * d + 0.5 * GRAVITY * d * d
    return out, states[-1]


class HipBackend:
    """The C ABI: device-resident front end, voxel map, ICP factors (batched re-linearization), photometric path."""

    def __init__(self, ctx, cfg: ReplayConfig, mode=synth.ENWIDE_NEIGHBOR_MODE):
        from . import capi
        self.capi, self.ctx, self.cfg = capi, ctx, cfg
        reg = cfg.reg
        self.reg = capi.make_reg_config(**reg)
        self.regd = reg
        self.map = capi.VoxelMap(ctx, leaf=reg["target_ivox_map_leaf_size"], min_dist=reg["target_ivox_map_min_dist_in_voxel"], mode=mode)
        self.scan = capi.Scan(ctx)
        self.icfg = capi.make_input_config()
        self.I3, self.z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        # the photometric path on its own context (stream): see host/mimosa_hip/replay.hpp
        self.photo_ctx = capi.Context(ctx.device) if cfg.photometric else None
        self.photo = capi.Photo(self.photo_ctx, cfg.photo) if cfg.photometric else None
        if self.photo is not None:
            ctx.check(ctx.L.mh_scan_keep_raw(self.scan.h, 1))

    def seed_map(self, xyz):
        self.map.insert(xyz)

    def prepare(self, raw):
        self.scan.prepare_input(raw, self.icfg)
        return self.scan.unique_ns()

    def deskew_and_preprocess(self, T_Le_Lt):
        self.scan.deskew(T_Le_Lt.astype(np.float32))
        if self.photo is not None:
            self.photo.preprocess_scan(self.scan, T_Le_Lt)
        info = self.scan.preprocess_geometric(self.I3, self.z3, self.regd["source_voxel_grid_filter_leaf_size"], 20,
                                              self.regd["source_voxel_grid_min_dist_in_voxel"])
        return info["n_downsampled"]

    def make_factor(self):
        f = self.scan.make_factor(self.map, self.reg)
        f.set_components(False)  # the loop takes H, b, f only (mh_icp_set_components)
        return f

    def make_photo_factor(self):
        return self.photo.make_factor() if self.photo is not None and self.photo.features() else None

    def linearize_window(self, factors, poses):
        """Every live ICP factor at its current pose: ONE batched call."""
        rs = self.capi.linearize_batch(factors, [p[0] for p in poses], [p[1] for p in poses])
        return [(np.asarray(r["H_ss"]).reshape(6, 6), np.asarray(r["b_s"]), float(r["f"])) for r in rs]

    def linearize_photo(self, pf, R, t):
        r = pf.linearize(R, t)
        return np.asarray(r["H_bb"]).reshape(6, 6), np.asarray(r["b_b"]), float(r["f"]), int(r["status_hist"][8])

    def update_map(self, R, t):
        # Geometric::updateMap: copy-then-insert (geometric.cpp:494-495), f32 world transform (:483-490) on the device
        new = self.map.copy()
        new.insert_from_scan(self.scan, R, t)
        self.map.release()
        self.map = new

    def photo_update_map(self, pf, R, t):
        self.photo.update_map(pf, R, t, synth_photo.BIAS_DIRECTIONS)

    def release(self, f):
        f.destroy()


def first_state(cfg: ReplayConfig, scan0, rng_seed=7):
    """State at the first IMU sample of the first sweep: ground truth + the prior error (the caller's first guess)."""
    rng = np.random.default_rng(rng_seed)
    Rs = scan0["R_gt"] @ synth.so3_exp(np.deg2rad(cfg.prior_rot_noise_deg) * rng.standard_normal(3))
    ps = scan0["t_gt"] + cfg.prior_trans_noise * rng.standard_normal(3)
    w_, v_body = np.asarray(cfg.w, float), np.asarray(cfg.v, float)
    R_start = Rs @ synth.so3_exp(-w_ * cfg.dt)
    p_start = ps - R_start @ (v_body * cfg.dt)
    return R_start, p_start, R_start @ v_body


def write_native_input(path, cfg: ReplayConfig, scans, rng_seed=7, mode=synth.ENWIDE_NEIGHBOR_MODE):
    """The input file of the native harness (mimosa_amd/host/replay_main.cpp): length-prefixed little-endian vectors."""
    import struct
    from . import capi
    with open(path, "wb") as f:
        def w(arr, dtype=None):
            arr = np.ascontiguousarray(arr if dtype is None else np.asarray(arr, dtype))
            f.write(struct.pack("<Q", len(arr) if arr.dtype.itemsize == 32 else arr.size))
            f.write(arr.tobytes())
        w([cfg.window, cfg.update_iters, int(cfg.photometric), mode, 1000], np.int32)
        w([cfg.between_sigma_rot, cfg.between_sigma_trans, cfg.keyframe_trans_thresh, cfg.keyframe_rot_thresh_deg, *GRAVITY], np.float64)
        w(np.frombuffer(bytes(capi.make_reg_config(**cfg.reg)), np.uint8))
        w(np.frombuffer(bytes(capi.make_input_config()), np.uint8))
        if cfg.photometric:
            pc = cfg.photo
            w([pc["rows"], pc["cols"], pc["destagger"], pc["erosion_buffer"], pc["patch_size"], pc["margin_size"], pc["remove_lines"],
               pc["filter_brightness"], pc["gaussian_blur"], pc["gaussian_blur_size"], pc["nma_radius"], pc["num_features_detect"],
               pc["max_feature_life_time"], pc["rotate_patch_to_align_with_gradient"], pc["use_robust_cost_function"],
               pc["robust_cost_function"], pc["brightness_window_size"][0], pc["brightness_window_size"][1]], np.int32)
            w([pc["range_min"], pc["range_max"], pc["intensity_scale"], pc["intensity_gamma"], pc["gradient_threshold"],
               pc["max_dist_from_mean"], pc["max_dist_from_plane"], pc["occlusion_range_diff_threshold"],
               pc["lidar_origin_to_beam_origin_mm"], pc["robust_cost_function_parameter"], pc["error_scale"], pc["max_error"],
               pc["sigma"]], np.float64)
            w(pc["pixel_shift_by_row"], np.int32)
            w(pc["beam_altitude_angles"], np.float32)
            w(pc["high_pass_fir"], np.float64)
            w(pc["low_pass_fir"], np.float64)
            w(np.asarray(pc["patch_offsets"], np.int32).ravel())
            w(np.concatenate([np.asarray(pc["T_B_L_R"], float).ravel(), np.asarray(pc["T_B_L_t"], float)]))
        w(np.asarray(synth_photo.BIAS_DIRECTIONS, np.float64).ravel())
        w(synth.make_room(synth.BASE_SEED, 0, 0, room=np.asarray(cfg.room)).astype(np.float32).ravel())
        R0, p0, v0 = first_state(cfg, scans[0], rng_seed)
        w(np.concatenate([R0.ravel(), p0, v0]))
        w([len(scans)], np.int32)
        for sc in scans:
            ts, gyro, acc = sc["imu"]
            w(sc["raw"])
            w(ts, np.float64)
            w(np.asarray(gyro, np.float64).ravel())
            w(np.asarray(acc, np.float64).ravel())
            w([sc["header_ts"]], np.float64)


def run_native(cfg: ReplayConfig, scans, workdir, repeats=1, rng_seed=7, visible_device=None, through_manager=False, sequential=False,
               sharded_world=0, sharded_rccl=False, timeout=900):
    """The same replay through the C++ host mirror (host/mimosa_hip/replay.hpp): no Python between the library calls.
    visible_device: run the driver with HIP_VISIBLE_DEVICES set to it (one replay per GPU of a node).
    through_manager: every scan goes through lidar::Manager::callback (host/mimosa_hip/manager.hpp: the reference's method
    names and call order, the first cloud initialises instead of being registered) instead of FixedLagReplay's own loop.
    sharded_world = W > 0: the map sharded over W ranks inside the driver process (host/mimosa_hip/sharded_replay.hpp, one host
    thread per rank, in-process transport); sharded_rccl: the driver is one rank of the launch this process belongs to (RANK /
    WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment), RCCL inside the library."""
    import json
    import os
    import subprocess
    import sys
    from . import build
    exe = build.build_replay_native()
    path = os.path.join(workdir, "replay_input.bin")
    write_native_input(path, cfg, scans, rng_seed)
    env = dict(os.environ)
    if visible_device is not None:
        env["HIP_VISIBLE_DEVICES"] = str(visible_device)
    mode = ["manager"] if through_manager else (["sequential"] if sequential else [])
    if sharded_world:
        mode = ["sharded", str(sharded_world)]
    if sharded_rccl:
        mode = ["sharded-rccl"]
    out = subprocess.run([exe, path, str(repeats)] + mode, capture_output=True, text=True, timeout=timeout, env=env)
    os.remove(path)
    if out.returncode != 0:
        raise RuntimeError("replay_native failed: " + out.stderr[-2000:])
    if out.stderr and (os.environ.get("MH_ALLOC_TRACE") or os.environ.get("MH_DETECT_TRACE")):
        sys.stderr.write(out.stderr)   # the library's diagnostic traces
    r = json.loads(out.stdout)
    r["poses_est"] = [(np.array(p[:9]).reshape(3, 3), np.array(p[9:])) for p in r.pop("poses")]
    return r


def run(cfg: ReplayConfig, backend, scans=None, rng_seed=7):
    """Replay: returns dict(poses_est, errors, keyframes, per-stage seconds, ...)."""
    scans = scans if scans is not None else make_scans(cfg)
    backend.seed_map(synth.make_room(synth.BASE_SEED, 0, 0, room=np.asarray(cfg.room)))
    stage = {"front_end": 0.0, "imu": 0.0, "factor_create": 0.0, "optimise": 0.0, "update_map": 0.0}
    v_body = np.asarray(cfg.v, float)
    Wb = np.diag([1.0 / cfg.between_sigma_rot**2] * 3 + [1.0 / cfg.between_sigma_trans**2] * 3)
    win = []          # live window: dicts(k, R, t, factor, Z (relative pose to the previous scan), fresh)
    est, kf_poses, n_kf, costs, n_photo_valid = [], [], 0, [], []
    R_prev = t_prev = vel_prev = None
    t0 = time.perf_counter()
    for k, sc in enumerate(scans):
        a0 = time.perf_counter()
        uns = backend.prepare(sc["raw"])
        a1 = time.perf_counter()
        # IMU propagation from the previous scan's estimate (first scan: ground truth + the prior error)
        if R_prev is None:
            R_start, p_start, vel_start = first_state(cfg, sc, rng_seed)   # state at the first IMU sample of this sweep
        else:
            R_start, p_start, vel_start = R_prev, t_prev, vel_prev
        T_W_Bt, (R_pred, p_pred, vel_pred) = propagate(R_start, p_start, vel_start, sc["imu"], sc["header_ts"], uns)
        T_Le_Lt = np.empty_like(T_W_Bt)
        T_Le_Lt[:, :9] = (R_pred.T @ T_W_Bt[:, :9].reshape(-1, 3, 3)).reshape(-1, 9)
        T_Le_Lt[:, 9:] = (T_W_Bt[:, 9:] - p_pred) @ R_pred
        a2 = time.perf_counter()
        backend.deskew_and_preprocess(T_Le_Lt)
        a3 = time.perf_counter()
        f = backend.make_factor()
        pf = backend.make_photo_factor() if cfg.photometric else None
        Z = _between(R_start, p_start, R_pred, p_pred) if R_prev is not None else None
        win.append(dict(k=k, R=R_pred, t=p_pred, f=f, Z=Z))
        if len(win) > cfg.window:
            backend.release(win.pop(0)["f"])
        a4 = time.perf_counter()
        # ---- smoother update: every live factor re-linearized per iteration -------------------------------------
        nW = len(win)
        fs = []
        for it in range(cfg.update_iters):
            lin = backend.linearize_window([w["f"] for w in win], [(w["R"], w["t"]) for w in win])
            A = np.zeros((6 * nW, 6 * nW))
            g = np.zeros(6 * nW)
            cost = 0.0
            for i, (H, b, fv) in enumerate(lin):
                A[6 * i:6 * i + 6, 6 * i:6 * i + 6] += H
                g[6 * i:6 * i + 6] += b
                cost += fv
            if pf is not None:
                Hp, bp, fp, nv = backend.linearize_photo(pf, win[-1]["R"], win[-1]["t"])
                if nv and np.all(np.isfinite(Hp)) and np.all(np.isfinite(bp)):
                    A[-6:, -6:] += Hp
                    g[-6:] += bp
                    cost += fp
            for i in range(1, nW):
                if win[i]["Z"] is None:
                    continue
                Rz, tz = win[i]["Z"]
                Rab, tab = _between(win[i - 1]["R"], win[i - 1]["t"], win[i]["R"], win[i]["t"])
                Re, te = Rz.T @ Rab, Rz.T @ (tab - tz)                     # Z^-1 * between(X_{i-1}, X_i)
                r = np.concatenate([_so3_log(Re), te])
                Jb = np.eye(6)
                Ja = -_adjoint(Rab.T, -Rab.T @ tab)                         # -Ad(between^-1)
                J = np.zeros((6, 6 * nW))
                J[:, 6 * (i - 1):6 * i] = Ja
                J[:, 6 * i:6 * i + 6] = Jb
                A += J.T @ Wb @ J
                g += J.T @ Wb @ r
                cost += float(r @ Wb @ r)
            # what marginalisation leaves on the oldest pose; loose while that pose has never been optimised
            loose = win[0]["k"] == 0 and k < cfg.window
            sr, st = (np.deg2rad(1.0), 0.1) if loose else (1e-4, 1e-4)
            A[:6, :6] += np.diag([1.0 / sr**2] * 3 + [1.0 / st**2] * 3)
            xi = np.linalg.solve(A + 1e-9 * np.eye(6 * nW), -g)
            for i, w in enumerate(win):
                w["R"], w["t"] = _retract(w["R"], w["t"], xi[6 * i:6 * i + 6])
            fs.append(cost)
        costs.append(fs)
        a5 = time.perf_counter()
        R, t = win[-1]["R"], win[-1]["t"]
        # ---- Geometric::updateMap's keyframe test (geometric.cpp:445-478): nearest stored pose by translation, then
        # the largest of |yaw|, |pitch|, |roll| of the relative rotation against the threshold
        is_kf = True
        if kf_poses:
            dists = [np.linalg.norm(t - tk) for _, tk in kf_poses]
            j = int(np.argmin(dists))
            d = kf_poses[j][0].T @ R
            ypr = (np.arctan2(d[1, 0], d[0, 0]), np.arctan2(-d[2, 0], np.hypot(d[2, 1], d[2, 2])), np.arctan2(d[2, 1], d[2, 2]))
            is_kf = dists[j] > cfg.keyframe_trans_thresh or max(abs(a) for a in ypr) > cfg.keyframe_rot_thresh_deg * 0.017453293
        if is_kf:
            backend.update_map(R, t)
            kf_poses.append((R, t))
            n_kf += 1
        if cfg.photometric:
            if pf is not None:
                n_photo_valid.append(int(pf.linearize(R, t)["status_hist"][8]))
            backend.photo_update_map(pf, R, t)
            if pf is not None:
                backend.release(pf)
        a6 = time.perf_counter()
        stage["front_end"] += (a1 - a0) + (a3 - a2)
        stage["imu"] += a2 - a1
        stage["factor_create"] += a4 - a3
        stage["optimise"] += a5 - a4
        stage["update_map"] += a6 - a5
        est.append((R, t))
        # velocity: the propagated one, carried into the corrected attitude
        R_prev, t_prev, vel_prev = R, t, R @ (R_pred.T @ vel_pred)
    for w in win:
        backend.release(w["f"])
    total = time.perf_counter() - t0
    terr = [float(np.linalg.norm(te - s["t_gt"])) for (_, te), s in zip(est, scans)]
    rerr = [float(np.rad2deg(np.linalg.norm(_so3_log(Re.T @ s["R_gt"])))) for (Re, _), s in zip(est, scans)]
    return {"poses_est": est, "trans_err": terr, "rot_err_deg": rerr, "n_keyframes": n_kf, "seconds": total,
            "scans_per_s": len(scans) / total, "stage_s": stage, "costs": costs, "photo_valid": n_photo_valid}
