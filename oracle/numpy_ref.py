"""ORACLE — TEST INFRASTRUCTURE ONLY.  Independent numpy restatement of the reference hot path.

Written straight from SURVEY.md Appendix A / B (and the reference lines cited below), with
deliberately different machinery from oracle/ref_cpu.hpp so that the two restatements check each
other: dict-of-lists voxel map, brute-force k-NN by stable argsort over the neighbour-voxel candidate
set, LAPACK `numpy.linalg.eigh` instead of the Eigen QR restatement.  Pure-Python loops: small
cases only.  PARITY UNPINNED (the reference has no tests; it cannot be built or imported here).

Only tests/ and tests/golden/make_golden.py import this.
"""
from __future__ import annotations

import numpy as np

# RejectStatus, include/mimosa/lidar/geometric_factor.hpp:35-46
UNPROCESSED, INSUFFICIENT, MAX_DIST, EIG_FAIL, MIN_EIG_LOW, LINE, PLANE_INVALID, MAX_ERROR, VALID = range(9)


def fast_floor(v: float) -> int:
    """include/mimosa/lidar/utils.hpp:218-222"""
    n = int(v)  # truncation toward zero
    return n - (1 if v < n else 0)


def neighbor_offsets(mode: int):
    """gtsam_points neighbor_offsets."""
    if mode == 1:
        return [(0, 0, 0)]
    if mode == 7:
        return [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    out = []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            for k in (-1, 0, 1):
                if mode == 19 and abs(i) + abs(j) + abs(k) == 3:
                    continue
                out.append((i, j, k))
    return out


class VoxelMap:
    """iVox semantics: first-come-first-kept with a min-distance rule and a per-voxel cap; LRU purge."""

    def __init__(self, leaf=0.5, min_dist=0.15, max_pts=20, mode=19, lru_horizon=100, lru_clear_cycle=10):
        self.inv_leaf = 1.0 / leaf
        self.min_sq = min_dist * min_dist
        self.max_pts = max_pts
        self.offsets = neighbor_offsets(mode)
        self.lru_horizon, self.lru_clear_cycle, self.lru_counter = lru_horizon, lru_clear_cycle, 0
        self.order = []   # voxel coords in creation order (flat_voxels)
        self.cells = {}   # coord -> [list of float64 xyz, lru]

    def coord(self, p):
        return tuple(fast_floor(float(c) * self.inv_leaf) for c in p)

    def insert(self, xyz_f32):
        for p in np.asarray(xyz_f32, dtype=np.float32).reshape(-1, 3):
            pd = p.astype(np.float64)
            c = self.coord(pd)
            cell = self.cells.get(c)
            if cell is None:
                cell = [[], self.lru_counter]
                self.cells[c] = cell
                self.order.append(c)
            cell[1] = self.lru_counter
            pts = cell[0]
            if len(pts) >= self.max_pts:
                continue
            if any(float(np.sum((q - pd) ** 2)) < self.min_sq for q in pts):
                continue
            pts.append(pd)
        self.lru_counter += 1
        if self.lru_counter % self.lru_clear_cycle == 0:
            keep = [c for c in self.order if not (self.cells[c][1] + self.lru_horizon < self.lru_counter)]
            self.cells = {c: self.cells[c] for c in keep}
            self.order = keep

    def candidates(self, q):
        c = self.coord(q)
        out = []
        for o in self.offsets:
            cell = self.cells.get((c[0] + o[0], c[1] + o[1], c[2] + o[2]))
            if cell is not None:
                out.extend(cell[0])
        return np.array(out).reshape(-1, 3)

    def knn(self, q, k):
        cand = self.candidates(q)
        if len(cand) == 0:
            return cand, np.zeros(0)
        d = np.sum((cand - q) ** 2, axis=1)
        order = np.argsort(d, kind="stable")[:k]  # stable = earlier-seen wins ties (strict '<' insertion)
        return cand[order], d[order]

    @property
    def num_points(self):
        return sum(len(v[0]) for v in self.cells.values())


def linearize(vmap: VoxelMap, pts_xyz_f32, cfg: dict, R_src, t_src, g_unit=(0, 0, -1.0),
              R_tgt=None, t_tgt=None, state=None):
    """ICPFactor::linearize, geometric_factor.hpp:231-562.

    `state` (dict of per-point arrays) persists the data-association cache across calls.
    Returns (result dict, state).
    """
    P = np.asarray(pts_xyz_f32, dtype=np.float32).astype(np.float64)
    N = len(P)
    R_src, t_src = np.asarray(R_src, float), np.asarray(t_src, float)
    binary = R_tgt is not None
    if binary:
        R_tgt, t_tgt = np.asarray(R_tgt, float), np.asarray(t_tgt, float)
        R = R_tgt.T @ R_src
        t = R_tgt.T @ (t_src - t_tgt)
    else:
        R, t = R_src, t_src
    if state is None:
        state = dict(q_da=np.zeros((N, 3)), mean=np.zeros((N, 3)), normal=np.zeros((N, 3)),
                     status=np.zeros(N, np.int32), count=0)
    state["count"] += 1
    k = int(cfg["num_corres_points"])
    sigma = float(np.float32(cfg["lidar_point_noise_std_dev"]))
    huber = float(np.float32(cfg["huber_threshold"]))
    da_thresh = float(np.float32(np.float32(cfg["target_ivox_map_min_dist_in_voxel"]) / np.float32(4)))
    max_d2 = float(np.float32(cfg["max_corres_distance"]) * np.float32(cfg["max_corres_distance"]))
    plane_valid = float(np.float32(cfg["plane_validity_distance"]))
    origin = t.copy()
    global_z = -np.asarray(g_unit, float)
    local_z = R.T @ global_z
    Pi = np.outer(local_z, local_z)

    H = np.zeros((6, 6)); Hst = np.zeros((6, 6)); Htt = np.zeros((6, 6))
    b = np.zeros(6); bt = np.zeros(6); f = 0.0
    Lrot = np.zeros((N, 3)); Ltrans = np.zeros((N, 3))
    e_rows = np.zeros(N); J_rows = np.zeros((N, 6))
    n_knn = 0; n_cand = 0
    for i in range(N):
        p = P[i]
        q = R @ p + t
        if np.linalg.norm(q - state["q_da"][i]) > da_thresh:
            state["q_da"][i] = q
            state["status"][i] = UNPROCESSED
            n_knn += 1
            n_cand += len(vmap.candidates(q))
            nb, d2 = vmap.knn(q, k)
            if len(nb) < k:
                state["status"][i] = INSUFFICIENT
                continue
            if d2[-1] > max_d2:
                state["status"][i] = MAX_DIST
                continue
            m = nb.mean(axis=0)
            state["mean"][i] = m
            Cc = nb - m
            cov = Cc.T @ Cc / (k - 1)
            lam, V = np.linalg.eigh(cov)
            if lam[0] < 1e-6:
                state["status"][i] = MIN_EIG_LOW
                continue
            if lam[2] > 3 * lam[1]:
                state["status"][i] = LINE
                continue
            n = V[:, 0].copy()
            if n @ (origin - m) < 0:
                n = -n
            state["normal"][i] = n
            if np.any(np.abs(Cc @ n) > plane_valid):
                state["status"][i] = PLANE_INVALID
                continue
        elif state["status"][i] <= PLANE_INVALID:
            continue
        m, n = state["mean"][i], state["normal"][i]
        e = float(n @ (m - q))
        s = 1 - 0.9 * abs(e) / np.sqrt(np.linalg.norm(p))
        if s < 0.9:
            state["status"][i] = MAX_ERROR
            continue
        sw = 1.0
        if cfg.get("use_huber", 1):
            w = e / sigma
            if abs(w) > huber:
                sw = np.sqrt(huber / abs(w))
        e *= sw / sigma
        ns = R.T @ n
        J = np.concatenate([np.cross(ns, p), -ns])
        nr = np.linalg.norm(J[:3])
        Lrot[i] = J[:3] / nr if nr > 0 else J[:3]
        Ltrans[i] = J[3:]
        J = J * (sw / sigma)
        H += np.outer(J, J); b += J * e; f += e * e
        e_rows[i] = e; J_rows[i] = J
        if binary:
            Jt = np.concatenate([np.cross(q, n), n]) * (sw / sigma)
            Hst += np.outer(J, Jt); Htt += np.outer(Jt, Jt); bt += Jt * e
        state["status"][i] = VALID

    def sqrt_eig(A):
        lam, V = np.linalg.eigh(A)
        with np.errstate(invalid="ignore"):
            return np.sqrt(lam), V

    loc_rot, E_rot = sqrt_eig(H[:3, :3])
    loc_trans, E_trans = sqrt_eig(H[3:, 3:])
    with np.errstate(all="ignore"):
        try:
            S_rr = np.linalg.inv(H[:3, :3] - H[:3, 3:] @ np.linalg.inv(H[3:, 3:]) @ H[3:, :3])
            S_tt = np.linalg.inv(H[3:, 3:] - H[3:, :3] @ np.linalg.inv(H[:3, :3]) @ H[:3, 3:])
            degen_rot, dE_rot = sqrt_eig(S_rr)
            degen_trans, dE_trans = sqrt_eig(S_tt)
            degen_rot = degen_rot * 57.29578  # RAD2DEG = PCL macro (x)*57.29578
        except np.linalg.LinAlgError:
            degen_rot = degen_trans = np.full(3, np.nan)
            dE_rot = dE_trans = np.full((3, 3), np.nan)
    valid = state["status"] == VALID
    tc = np.abs(Ltrans[valid] @ E_trans); tc[tc < 0.5] = 0
    rc = np.abs(Lrot[valid] @ E_rot); rc[rc < 0.5] = 0
    res = dict(loc_trans_comp=tc.sum(axis=0), loc_rot_comp=rc.sum(axis=0),
               loc_trans_final=loc_trans, loc_rot_final=loc_rot, eigvec_trans=E_trans, eigvec_rot=E_rot,
               degen_rot=degen_rot, degen_trans=degen_trans, degen_eigvec_rot=dE_rot,
               degen_eigvec_trans=dE_trans)
    if not binary:
        if cfg.get("reg_4_dof", 0):
            H = H.copy()
            H[:3, :3] = Pi @ H[:3, :3] @ Pi
            H[:3, 3:] = Pi @ H[:3, 3:]
            H[3:, :3] = H[3:, :3] @ Pi
            b = b.copy(); b[:3] = Pi @ b[:3]
        if cfg.get("project_on_degneneracy", 0):
            rot_degen = not np.all(loc_rot > float(np.float32(cfg["degen_thresh_rot"])))
            trans_degen = not np.all(loc_trans > float(np.float32(cfg["degen_thresh_trans"])))
            if rot_degen or trans_degen:
                # reference quirk F10: rebuilt from never-written zero arrays (:270-271, :496-532)
                H = np.zeros((6, 6)); b = np.zeros(6)
                res["loc_rot_final"], res["eigvec_rot"] = sqrt_eig(np.zeros((3, 3)))
                res["loc_trans_final"], res["eigvec_trans"] = sqrt_eig(np.zeros((3, 3)))
    res.update(H_ss=H, H_st=Hst, H_tt=Htt, b_s=b, b_t=bt, f=f,
               status_hist=np.bincount(state["status"], minlength=9).astype(np.int32),
               linearize_count=state["count"], n_knn=n_knn,
               mean_candidates=(n_cand / n_knn if n_knn else 0.0), e_rows=e_rows, J_rows=J_rows)
    return res, state


def transform_f32(xyz, R, t):
    """f32 R*p + t in the reference's operation order, no FMA: r0*x + (r1*y + r2*z), then + t
    (src/lidar/manager.cpp:504-508, src/lidar/geometric.cpp:154-161,483-490)."""
    xyz = np.asarray(xyz, np.float32)
    R = np.asarray(R, np.float32).reshape(3, 3)
    t = np.asarray(t, np.float32)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out = np.empty_like(xyz)
    for r in range(3):
        out[:, r] = (R[r, 0] * x + (R[r, 1] * y + R[r, 2] * z)) + t[r]
    return out


def deskew(xyz, t_ns, unique_ns, Rt12):
    """Manager::deskewPoints hot loop (src/lidar/manager.cpp:496-509)."""
    xyz = np.asarray(xyz, np.float32).copy()
    unique_ns = np.asarray(unique_ns, np.uint32)
    g = np.searchsorted(unique_ns, t_ns)
    for gi in range(len(unique_ns)):
        sel = (g == gi) & (np.asarray(t_ns) == unique_ns[gi])
        if sel.any():
            xyz[sel] = transform_f32(xyz[sel], Rt12[gi, :9], Rt12[gi, 9:])
    return xyz


def downsample(xyz_f32, leaf=0.5, max_pts=20, min_dist=0.15):
    """Geometric::downsample (src/lidar/geometric.cpp:55-126): kept indices, voxels in first-seen
    order, points in acceptance order."""
    inv = 1.0 / leaf
    min_sq = min_dist * min_dist
    order, cells = [], {}
    P = np.asarray(xyz_f32, np.float32).astype(np.float64)
    for i, p in enumerate(P):
        c = tuple(fast_floor(float(v) * inv) for v in p)
        cell = cells.get(c)
        if cell is None:
            cell = ([], [])
            cells[c] = cell
            order.append(c)
        pts, idx = cell
        if len(pts) >= max_pts:
            continue
        if any(float(np.sum((q - p) ** 2)) < min_sq for q in pts):
            continue
        pts.append(p); idx.append(i)
    return np.array([i for c in order for i in cells[c][1]], dtype=np.uint32)


def prepare_input(raw, range_min=0.2, range_max=100.0, intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9,
                  z_offset=0.0, create_full_res_pointcloud=True, point_skip_divisor=4, ring_skip_divisor=1):
    """Manager::prepareInput<PointOuster> (src/lidar/manager.cpp:244-368), vectorised: boolean masks instead of
    the reference's sequential `continue` chain, numpy float32 arithmetic for the range test.
    raw: structured array with x, y, z, intensity (f4), t (u4), ring (u2).  Returns points_full fields as a
    dict of arrays, geometric indices (into points_full), unique_ns, groups, last_point_ns."""
    n = len(raw)
    i = np.arange(n)
    x, y, z = raw["x"].astype(np.float32), raw["y"].astype(np.float32), raw["z"].astype(np.float32)
    inten = raw["intensity"].astype(np.float32)
    stride = 1 if create_full_res_pointcloud else point_skip_divisor
    with np.errstate(invalid="ignore"):
        keep = (i % stride) == 0
        keep &= ~(np.isnan(x) | np.isnan(y) | np.isnan(z))                                            # :253
        keep &= ~(np.isnan(inten) | (inten < np.float32(intensity_min)) | (inten > np.float32(intensity_max)))  # :272-276
        r2 = (x * x + y * y) + z * z                                                                    # :281, f32, left to right
        rmin2 = np.float32(range_min) * np.float32(range_min)
        rmax2 = np.float32(range_max) * np.float32(range_max)
        keep &= ~((r2 < rmin2) | (r2 > rmax2))                                                          # :282
        keep &= ~(raw["t"].astype(np.float32) > np.float32(ns_max))                                     # :306
    sel = np.nonzero(keep)[0]
    full = {
        "x": x[sel], "y": y[sel], "z": z[sel] + np.float32(z_offset), "intensity": inten[sel], "t": raw["t"][sel],
        "idx": sel.astype(np.uint32), "range": np.sqrt(r2[sel]),
    }
    geo_mask = ((sel % point_skip_divisor) == 0) & ((raw["ring"][sel] % ring_skip_divisor) == 0)     # :318, :331
    geo = np.nonzero(geo_mask)[0]
    t_kept = raw["t"][sel]
    unique_ns = np.unique(t_kept)                                                                       # :340-368
    groups = [np.nonzero(t_kept == u)[0] for u in unique_ns]
    last = int(t_kept.max()) if len(sel) else 0
    return {"points_full": full, "geometric_idxs": geo, "unique_ns": unique_ns, "groups": groups, "last_point_ns": last}


def prepare_input_typed(kind, raw, header_ts=0.0, width=None, height=1, transpose=False, organize=False, range_min=0.2, range_max=100.0,
                        intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9, z_offset=0.0, create_full_res_pointcloud=True,
                        point_skip_divisor=4, ring_skip_divisor=1):
    """Manager::prepareInput<PointT> (src/lidar/manager.cpp:149-383) for the reference's nine point types, vectorised and
    independent of oracle/ref_cpu.hpp: re-ordering by index arithmetic (transpose :177-203, stable bucket-by-ring :205-241),
    masks instead of the `continue` chain, per-type time decoding (:285-304) in numpy float64 / float32.
    raw: structured array with the type's own field names (x, y, z, intensity | reflectivity, t | timestamp | time, ring, tag)."""
    n = len(raw)
    width = n if width is None else width
    order = np.arange(n)
    if transpose and kind in ("rslidar", "velodyne_anybotics"):
        # transposed[new_row * new_width + new_col] = cloud[new_col * width + new_row], new_width = height
        j = np.arange(n)
        order = (j % height) * width + (j // height)
        width, height = height, width
    raw = raw[order]
    has_ring = kind not in ("livox", "livox_custom2", "ouster_odyssey")
    if organize and height == 1 and has_ring:
        raw = raw[np.argsort(raw["ring"].astype(np.int64), kind="stable")]
    i = np.arange(n)
    x, y, z = raw["x"].astype(np.float32), raw["y"].astype(np.float32), raw["z"].astype(np.float32)
    stride = 1 if create_full_res_pointcloud else point_skip_divisor
    with np.errstate(invalid="ignore", over="ignore"):
        keep = (i % stride) == 0
        keep &= ~(np.isnan(x) | np.isnan(y) | np.isnan(z))
        if kind in ("livox", "livox_custom2"):                                         # :256-262
            keep &= np.isin(raw["tag"] & 0x30, (0x00, 0x10))
        if kind == "ouster_odyssey":                                                   # :265-271
            refl = raw["reflectivity"]
            keep &= ~((refl < np.float32(intensity_min)) | (refl > np.float32(intensity_max)))
            inten = refl.astype(np.float32)
        else:
            inten = raw["intensity"].astype(np.float32)
            keep &= ~(np.isnan(inten) | (inten < np.float32(intensity_min)) | (inten > np.float32(intensity_max)))
        r2 = (x * x + y * y) + z * z
        keep &= ~((r2 < np.float32(range_min) * np.float32(range_min)) | (r2 > np.float32(range_max) * np.float32(range_max)))
        if kind in ("ouster", "ouster_odyssey", "ouster_r8", "livox_custom2"):
            t_ns = raw["t"].astype(np.uint32)
        else:
            if kind in ("hesai", "rslidar"):
                d = (raw["timestamp"].astype(np.float64) - np.float64(header_ts)) * 1e9
            elif kind == "livox":
                d = raw["timestamp"].astype(np.float64) - np.float64(header_ts) * 1e9
            else:
                d = raw["time"].astype(np.float32).astype(np.float64) * 1e9
            # uint32_t t_ns = <double> on x86-64: truncate to a 64-bit integer, keep the low word
            t_ns = (np.trunc(d).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
        keep &= ~(t_ns.astype(np.float32) > np.float32(ns_max))
    sel = np.nonzero(keep)[0]
    full = {"x": x[sel], "y": y[sel], "z": z[sel] + np.float32(z_offset), "intensity": inten[sel], "t": t_ns[sel],
            "idx": sel.astype(np.uint32), "range": np.sqrt(r2[sel])}
    geo_mask = (sel % point_skip_divisor) == 0
    if kind not in ("livox", "livox_custom2", "velodyne_anybotics", "ouster_odyssey"):  # :321-332
        geo_mask &= (raw["ring"][sel].astype(np.int64) % ring_skip_divisor) == 0
    unique_ns = np.unique(t_ns[sel])
    return {"points_full": full, "geometric_idxs": np.nonzero(geo_mask)[0], "unique_ns": unique_ns,
            "last_point_ns": int(t_ns[sel].max()) if len(sel) else 0}


def _so3_exp(w):
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def deskew_poses(imu_t, imu_acc, imu_gyro, nav_R, nav_p, nav_v, bias_acc, bias_gyro, g_unit, g_norm, unique_ns,
                 header_ts, T_B_S):
    """Manager::deskewPoints pose part (src/lidar/manager.cpp:455-499) with 4x4 homogeneous matrices: for each
    distinct timestamp inside IMU interval [t_c, t_c+1]: R = R_c Exp(omega dt), p = p_c + v_c dt + 1/2 R_c a dt^2
    + 1/2 g |g| dt^2; T_Le_Lt = T_B_S^-1 T_W_Be^-1 T_W_Bt T_B_S."""
    def hom(R, p):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, p
        return T
    T_BS = hom(*T_B_S)
    T_W_Be = hom(nav_R[-1], nav_p[-1])
    out, u = [], 0
    for c in range(len(imu_t) - 1):
        while u < len(unique_ns):
            ts = header_ts + float(unique_ns[u]) * 1.0e-9
            if ts > imu_t[c + 1]:
                break
            dt = ts - imu_t[c]
            a, w = np.asarray(imu_acc[c]) - bias_acc, np.asarray(imu_gyro[c]) - bias_gyro
            R = nav_R[c] @ _so3_exp(w * dt)
            p = nav_p[c] + nav_v[c] * dt + 0.5 * nav_R[c] @ a * dt * dt + 0.5 * np.asarray(g_unit) * g_norm * dt * dt
            out.append(np.linalg.inv(T_BS) @ np.linalg.inv(T_W_Be) @ hom(R, p) @ T_BS)
            u += 1
    return out
