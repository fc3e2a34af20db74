"""Sequence replay harness (SURVEY.md §8 "next" row f-4, BASELINE configs[4] minus the photometric factor).

A synthetic platform drives through the rooms world at a constant body twist.  Every scan goes through the
reference's LiDAR call order (src/lidar/manager.cpp:45-147):

    prepareInput -> deskewPoints -> Geometric::preprocess -> getFactors (ICPFactor ctor + first linearize)
    -> optimiser re-linearizations -> Geometric::updateMap (keyframe test, copy-then-insert)

The optimiser is a plain Gauss-Newton on the single 6x6 factor (GTSAM / ISAM2 stay out of scope, SURVEY.md
§2): solve H xi = -b, retract T <- T Exp(xi) with xi = (omega, v) in the body frame — the perturbation the
reference's Jacobian J = [(n_s x p)^T, -n_s^T] is taken against (geometric_factor.hpp:341-355).  The motion
prior is constant velocity; the per-timestamp deskew poses come from the same constant-twist model the
reference's IMU extrapolation reduces to when acceleration is zero (manager.cpp:478-489).

`backend` is the object that does the numerical work:  HipBackend drives the C ABI (device-resident scan
front end + ICP factor);  tests/ build the same loop on the CPU oracle to check the trajectory.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import synth


@dataclass
class ReplayConfig:
    n_scans: int = 20
    rows: int = 128
    cols: int = 1024
    dt: float = 0.1                                   # 10 Hz
    v: tuple = (2.0, 0.3, 0.0)                        # body-frame twist of the platform
    w: tuple = (0.0, 0.0, 0.5)
    start_local: tuple = (20.0, 20.0, 1.7)
    start_yaw: float = 0.3
    room: tuple = tuple(synth.ROOM)
    gn_iters: int = 3
    prior_trans_noise: float = 0.03                   # initial-guess error fed to the optimiser, metres
    prior_rot_noise_deg: float = 0.3
    keyframe_trans_thresh: float = 1.0                # ENWIDE: map_keyframe_trans_thresh
    keyframe_rot_thresh_deg: float = 20.0
    seed_map_with_room: bool = True                   # start from a surveyed room instead of an empty map
    reg: dict = field(default_factory=synth.enwide_config)


def _exp_se3_right(R, t, xi):
    """T <- T * Exp(xi), xi = (omega, v), first-order in the translation (gtsam Pose3::Retract)."""
    R2 = R @ synth.so3_exp(np.asarray(xi[:3]))
    return R2, t + R @ np.asarray(xi[3:])


def trajectory(cfg: ReplayConfig):
    """Ground-truth scan-END sensor poses: constant body twist integrated exactly per scan."""
    R = synth.rot_z(cfg.start_yaw)
    t = synth.room_origin(0, 0) + np.asarray(cfg.start_local, float)
    v, w = np.asarray(cfg.v, float), np.asarray(cfg.w, float)
    out = []
    for _ in range(cfg.n_scans):
        out.append((R.copy(), t.copy()))
        R, t = R @ synth.so3_exp(w * cfg.dt), t + R @ (v * cfg.dt)
    return out


def make_scans(cfg: ReplayConfig):
    """Raw Ouster clouds along the trajectory + their exact per-column deskew poses."""
    scans = []
    for k, (R, t) in enumerate(trajectory(cfg)):
        yaw = float(np.arctan2(R[1, 0], R[0, 0]))
        raw, aux = synth.make_raw_scan(cfg.rows, seed=synth.BASE_SEED + 100 + k, n_cols=cfg.cols, room=np.asarray(cfg.room),
                                       sensor_local=t - synth.room_origin(0, 0), v=cfg.v, w=cfg.w, yaw=yaw)
        scans.append((raw, aux, R, t))
    return scans


class HipBackend:
    """The C ABI: device-resident front end, ICP factor, voxel map."""

    def __init__(self, ctx, reg: dict, mode=synth.ENWIDE_NEIGHBOR_MODE):
        from . import capi
        self.capi, self.ctx = capi, ctx
        self.reg = capi.make_reg_config(**reg)
        self.regd = reg
        self.map = capi.VoxelMap(ctx, leaf=reg["target_ivox_map_leaf_size"], min_dist=reg["target_ivox_map_min_dist_in_voxel"],
                                 mode=mode)
        self.scan = capi.Scan(ctx)
        self.icfg = capi.make_input_config()
        self.I3, self.z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)

    def seed_map(self, xyz):
        self.map.insert(xyz)

    def front_end(self, raw, aux):
        info = self.scan.prepare_input(raw, self.icfg)
        uns = self.scan.unique_ns()
        self.scan.deskew(aux["Rt12"][np.searchsorted(aux["unique_ns"], uns)])  # the pose of each kept timestamp
        info = self.scan.preprocess_geometric(self.I3, self.z3, self.regd["source_voxel_grid_filter_leaf_size"], 20,
                                              self.regd["source_voxel_grid_min_dist_in_voxel"])
        return info["n_downsampled"]

    def make_factor(self):
        return self.scan.make_factor(self.map, self.reg)

    def linearize(self, f, R, t):
        r = f.linearize(R, t)
        return np.asarray(r["H_ss"]).reshape(6, 6), np.asarray(r["b_s"]), float(r["f"]), r

    def body_cloud_xyz(self):
        b = self.scan.points(self.capi.Scan.BODY)
        return b

    def update_map(self, body, R, t):
        # Geometric::updateMap: copy-then-insert (geometric.cpp:494-495) with the f32 world transform (:483-490) applied
        # on the device to the resident Be_cloud_: nothing crosses PCIe, the old map stays valid for the factors holding it
        new = self.map.copy()
        new.insert_from_scan(self.scan, R, t)
        self.map.release()
        self.map = new


def run(cfg: ReplayConfig, backend, scans=None, rng_seed=7):
    """Replay: returns dict(poses_est, poses_gt, times, errors, keyframes)."""
    scans = scans if scans is not None else make_scans(cfg)
    rng = np.random.default_rng(rng_seed)
    if cfg.seed_map_with_room:
        backend.seed_map(synth.make_room(synth.BASE_SEED, 0, 0, room=np.asarray(cfg.room)))
    est, stage = [], {"front_end": 0.0, "factor_create": 0.0, "optimise": 0.0, "update_map": 0.0}
    kf_poses, n_kf, costs = [], 0, []
    R_prev = t_prev = R_prev2 = t_prev2 = None
    t0 = time.perf_counter()
    for k, (raw, aux, R_gt, t_gt) in enumerate(scans):
        a = time.perf_counter()
        n_ds = backend.front_end(raw, aux)
        b = time.perf_counter()
        # motion prior: constant velocity from the last two estimates (first scans: ground truth + noise)
        if R_prev2 is not None:
            dR, dt_ = R_prev2.T @ R_prev, R_prev2.T @ (t_prev - t_prev2)
            R0, t0_ = R_prev @ dR, t_prev + R_prev @ dt_
        else:
            R0, t0_ = R_gt, t_gt
        R0 = R0 @ synth.so3_exp(np.deg2rad(cfg.prior_rot_noise_deg) * rng.standard_normal(3))
        t0_ = t0_ + cfg.prior_trans_noise * rng.standard_normal(3)
        if cfg.seed_map_with_room or k > 0:
            f = backend.make_factor()
            c = time.perf_counter()
            R, t = R0, t0_
            fs = []
            for _ in range(cfg.gn_iters):
                H, bvec, fval, _ = backend.linearize(f, R, t)
                fs.append(fval)
                xi = np.linalg.solve(H + 1e-9 * np.eye(6), -bvec)
                R, t = _exp_se3_right(R, t, xi)
            costs.append(fs)
            if hasattr(f, "destroy"):
                f.destroy()
        else:
            c = time.perf_counter()
            R, t = R_gt, t_gt  # very first scan of an empty map defines the frame
        d = time.perf_counter()
        # keyframe test (geometric.cpp:445-478): nearest stored pose by translation, then rotation
        is_kf = True
        if kf_poses:
            dists = [np.linalg.norm(t - tk) for _, tk in kf_poses]
            j = int(np.argmin(dists))
            ang = np.rad2deg(np.arccos(np.clip((np.trace(kf_poses[j][0].T @ R) - 1) / 2, -1, 1)))
            is_kf = dists[j] > cfg.keyframe_trans_thresh or ang > cfg.keyframe_rot_thresh_deg
        if is_kf:
            backend.update_map(backend.body_cloud_xyz(), R, t)
            kf_poses.append((R, t))
            n_kf += 1
        e = time.perf_counter()
        stage["front_end"] += b - a
        stage["factor_create"] += c - b
        stage["optimise"] += d - c
        stage["update_map"] += e - d
        est.append((R, t))
        R_prev2, t_prev2, R_prev, t_prev = R_prev, t_prev, R, t
    total = time.perf_counter() - t0
    terr = [float(np.linalg.norm(te - tg)) for (_, te), (_, _, _, tg) in zip(est, scans)]
    rerr = [float(np.rad2deg(np.arccos(np.clip((np.trace(Re.T @ Rg) - 1) / 2, -1, 1)))) for (Re, _), (_, _, Rg, _) in zip(est, scans)]
    return {"poses_est": est, "trans_err": terr, "rot_err_deg": rerr, "n_keyframes": n_kf, "seconds": total,
            "scans_per_s": len(scans) / total, "stage_s": stage, "costs": costs}
