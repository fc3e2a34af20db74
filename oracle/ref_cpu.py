"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding over oracle/libref_cpu.so (the C++ CPU
restatement of the reference hot path, see ref_cpu.hpp).  PARITY UNPINNED (no reference tests
exist).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RegistrationConfig(C.Structure):
    # include/mimosa/lidar/geometric_config.hpp:17-33
    _fields_ = [
        ("source_voxel_grid_filter_leaf_size", C.c_float),
        ("source_voxel_grid_min_dist_in_voxel", C.c_float),
        ("target_ivox_map_leaf_size", C.c_float),
        ("target_ivox_map_min_dist_in_voxel", C.c_float),
        ("num_corres_points", C.c_uint64),
        ("max_corres_distance", C.c_float),
        ("plane_validity_distance", C.c_float),
        ("lidar_point_noise_std_dev", C.c_float),
        ("use_huber", C.c_int32),
        ("huber_threshold", C.c_float),
        ("reg_4_dof", C.c_int32),
        ("project_on_degneneracy", C.c_int32),
        ("degen_thresh_rot", C.c_float),
        ("degen_thresh_trans", C.c_float),
    ]


class LinearizeResult(C.Structure):
    _fields_ = [
        ("H_ss", C.c_double * 36), ("H_st", C.c_double * 36), ("H_tt", C.c_double * 36),
        ("b_s", C.c_double * 6), ("b_t", C.c_double * 6), ("f", C.c_double),
        ("loc_trans_comp", C.c_double * 3), ("loc_rot_comp", C.c_double * 3),
        ("loc_trans_final", C.c_double * 3), ("loc_rot_final", C.c_double * 3),
        ("eigvec_trans", C.c_double * 9), ("eigvec_rot", C.c_double * 9),
        ("degen_rot", C.c_double * 3), ("degen_trans", C.c_double * 3),
        ("degen_eigvec_rot", C.c_double * 9), ("degen_eigvec_trans", C.c_double * 9),
        ("status_hist", C.c_int32 * 9), ("linearize_count", C.c_int32),
        ("mean_candidates", C.c_double), ("n_knn", C.c_int64),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = np.array(v) if hasattr(v, "__len__") else v
        for k in ("H_ss", "H_st", "H_tt"):
            d[k] = d[k].reshape(6, 6)
        for k in ("eigvec_trans", "eigvec_rot", "degen_eigvec_rot", "degen_eigvec_trans"):
            d[k] = d[k].reshape(3, 3)
        return d


def make_config(**kw) -> RegistrationConfig:
    c = RegistrationConfig()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libref_cpu.so")
    srcs = [os.path.join(_HERE, f) for f in ("ref_cpu_capi.cpp", "ref_cpu.hpp", "photo_ref_capi.cpp", "photo_ref.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libref_cpu.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.ref_map_create.restype = vp
        L.ref_map_create.argtypes = [dbl, dbl, C.c_int, C.c_int, C.c_int]
        L.ref_map_set_lru_clear_cycle.argtypes = [vp, C.c_int]
        L.ref_map_copy.restype = vp
        L.ref_map_copy.argtypes = [vp]
        L.ref_map_destroy.argtypes = [vp]
        L.ref_map_insert.argtypes = [vp, vp, i64]
        L.ref_map_num_voxels.restype = i64
        L.ref_map_num_voxels.argtypes = [vp]
        L.ref_map_num_points.restype = i64
        L.ref_map_num_points.argtypes = [vp]
        L.ref_map_export.argtypes = [vp, vp, vp, vp]
        L.ref_map_knn.argtypes = [vp, vp, i64, C.c_int, vp, vp, vp, vp]
        L.ref_map_point.argtypes = [vp, i64, vp]
        L.ref_icp_create.restype = vp
        L.ref_icp_create.argtypes = [C.c_int, vp, vp, i64, C.POINTER(RegistrationConfig)]
        L.ref_icp_clone.restype = vp
        L.ref_icp_clone.argtypes = [vp]
        L.ref_icp_destroy.argtypes = [vp]
        L.ref_icp_set_threads.argtypes = [vp, C.c_int]
        L.ref_icp_linearize.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(LinearizeResult)]
        L.ref_icp_get_state.argtypes = [vp, vp, vp, vp, vp]
        L.ref_icp_get_da.argtypes = [vp, vp]
        L.ref_icp_set_state.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.ref_icp_point_rows.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ref_icp_time_cold.argtypes = [vp, vp, i64, C.POINTER(RegistrationConfig), vp, vp, vp, C.c_int,
                                        C.c_int, vp, C.POINTER(LinearizeResult)]
        L.ref_eigen3.argtypes = [vp, vp, vp]
        L.ref_deskew.argtypes = [vp, i64, vp, vp, i64]
        L.ref_transform_f32.argtypes = [vp, i64, vp, vp]
        L.ref_downsample.restype = i64
        L.ref_downsample.argtypes = [vp, i64, dbl, C.c_int, dbl, vp]
        L.ref_projection_matrix.argtypes = [vp, dbl, vp, vp, vp]
        L.ref_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Map:
    def __init__(self, leaf=0.5, min_dist=0.15, max_pts=20, mode=19, lru_horizon=1000, _h=None):
        self.L = lib()
        self.h = _h if _h is not None else self.L.ref_map_create(leaf, min_dist, max_pts, mode, lru_horizon)

    def copy(self):
        return Map(_h=self.L.ref_map_copy(self.h))

    def set_lru_clear_cycle(self, cycle: int):
        self.L.ref_map_set_lru_clear_cycle(self.h, int(cycle))

    def insert(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.L.ref_map_insert(self.h, _p(xyz), xyz.shape[0])

    @property
    def num_voxels(self):
        return self.L.ref_map_num_voxels(self.h)

    @property
    def num_points(self):
        return self.L.ref_map_num_points(self.h)

    def export(self):
        nv, npt = self.num_voxels, self.num_points
        coords = np.empty((nv, 3), np.int32)
        counts = np.empty(nv, np.int32)
        xyz = np.empty((npt, 3), np.float32)
        self.L.ref_map_export(self.h, _p(coords), _p(counts), _p(xyz))
        return coords, counts, xyz

    def knn(self, q, k=5):
        q = _f64(q).reshape(-1, 3)
        n = q.shape[0]
        idx = np.empty((n, k), np.int64)
        sq = np.empty((n, k), np.float64)
        found = np.empty(n, np.int32)
        ncand = np.empty(n, np.int32)
        self.L.ref_map_knn(self.h, _p(q), n, k, _p(idx), _p(sq), _p(found), _p(ncand))
        return idx, sq, found, ncand

    def point(self, gid):
        out = np.empty(3)
        self.L.ref_map_point(self.h, int(gid), _p(out))
        return out

    def __del__(self):
        try:
            self.L.ref_map_destroy(self.h)
        except Exception:
            pass


class ICP:
    def __init__(self, map_: Map, pts, cfg: RegistrationConfig, binary=False, _h=None):
        self.L = lib()
        self.map = map_
        self.n = len(pts) if pts is not None else None
        if _h is not None:
            self.h = _h
        else:
            pts = np.ascontiguousarray(pts)
            assert pts.dtype.itemsize == 32
            self.h = self.L.ref_icp_create(int(binary), map_.h, _p(pts), len(pts), C.byref(cfg))

    def clone(self):
        c = ICP(self.map, None, None, _h=self.L.ref_icp_clone(self.h))
        c.n = self.n
        return c

    def set_threads(self, n):
        self.L.ref_icp_set_threads(self.h, n)

    def linearize(self, R, t, g_unit=(0.0, 0.0, -1.0), R_tgt=None, t_tgt=None):
        out = LinearizeResult()
        R, t, g = _f64(R), _f64(t), _f64(g_unit)
        Rt = _f64(R_tgt) if R_tgt is not None else None
        tt = _f64(t_tgt) if t_tgt is not None else None
        self.L.ref_icp_linearize(self.h, _p(R), _p(t), _p(Rt), _p(tt), _p(g), C.byref(out))
        return out.as_dict()

    def state(self):
        st = np.empty(self.n, np.int32)
        means = np.empty((self.n, 3))
        normals = np.empty((self.n, 3))
        transed = np.empty((self.n, 3))
        self.L.ref_icp_get_state(self.h, _p(st), _p(means), _p(normals), _p(transed))
        return st, means, normals, transed

    def da_state(self):
        """(status, mean, normal, q_da): everything a point's data association consists of."""
        st, means, normals, _ = self.state()
        q = np.empty((self.n, 3))
        self.L.ref_icp_get_da(self.h, _p(q))
        return st, means, normals, q

    def set_da_state(self, st, means, normals, q_da, linearize_count=0):
        st = np.ascontiguousarray(st, np.int32)
        means, normals, q_da = _f64(means), _f64(normals), _f64(q_da)
        self.L.ref_icp_set_state(self.h, _p(st), _p(means), _p(normals), _p(q_da), int(linearize_count))

    def point_rows(self, R, t):
        e = np.empty(self.n)
        J = np.empty((self.n, 6))
        valid = np.empty(self.n, np.int32)
        R, t = _f64(R), _f64(t)
        self.L.ref_icp_point_rows(self.h, _p(R), _p(t), _p(e), _p(J), _p(valid))
        return e, J, valid

    def __del__(self):
        try:
            self.L.ref_icp_destroy(self.h)
        except Exception:
            pass


def time_cold(map_: Map, pts, cfg, R, t, g_unit=(0.0, 0.0, -1.0), n_threads=4, iters=3):
    L = lib()
    pts = np.ascontiguousarray(pts)
    secs = np.empty(iters)
    out = LinearizeResult()
    R, t, g = _f64(R), _f64(t), _f64(g_unit)
    L.ref_icp_time_cold(map_.h, _p(pts), len(pts), C.byref(cfg), _p(R), _p(t), _p(g), n_threads, iters,
                        _p(secs), C.byref(out))
    return secs, out.as_dict()


def eigen3(A):
    A = _f64(A)
    ev = np.empty(3)
    V = np.empty((3, 3))
    ok = lib().ref_eigen3(_p(A), _p(ev), _p(V))
    return bool(ok), ev, V


def deskew(pts, unique_ns, Rt12):
    pts = np.ascontiguousarray(pts).copy()
    u = np.ascontiguousarray(unique_ns, dtype=np.uint32)
    P = np.ascontiguousarray(Rt12, dtype=np.float32)
    lib().ref_deskew(_p(pts), len(pts), _p(u), _p(P), len(u))
    return pts


def transform_f32(pts, R, t):
    pts = np.ascontiguousarray(pts).copy()
    R = np.ascontiguousarray(R, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    lib().ref_transform_f32(_p(pts), len(pts), _p(R), _p(t))
    return pts


class InputConfig(C.Structure):
    _fields_ = [
        ("range_min", C.c_float), ("range_max", C.c_float), ("intensity_min", C.c_float), ("intensity_max", C.c_float),
        ("ns_max", C.c_float), ("z_offset", C.c_float), ("create_full_res_pointcloud", C.c_int32),
        ("point_skip_divisor", C.c_int32), ("ring_skip_divisor", C.c_int32),
    ]


def make_input_config(**kw) -> InputConfig:
    d = dict(range_min=0.2, range_max=100.0, intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9, z_offset=0.0,
             create_full_res_pointcloud=1, point_skip_divisor=4, ring_skip_divisor=1)
    d.update(kw)
    c = InputConfig()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def prepare_input(raw, cfg: InputConfig):
    """Manager::prepareInput<PointOuster> (src/lidar/manager.cpp:149-383).  Returns a dict: points_full
    (32-byte records), geometric_idxs, unique_ns, groups (list of index arrays), last_point_ns."""
    L = lib()
    raw = np.ascontiguousarray(raw)
    assert raw.dtype.itemsize == 32
    n = len(raw)
    full = np.zeros(max(n, 1), dtype=np.dtype((np.void, 32)))
    geo = np.zeros(max(n, 1), np.uint64)
    uniq = np.zeros(max(n, 1), np.uint32)
    goff = np.zeros(n + 2, np.uint64)
    gidx = np.zeros(max(n, 1), np.uint64)
    counts = np.zeros(4, np.uint64)
    L.ref_prepare_input.restype = C.c_int64
    L.ref_prepare_input(_p(raw), C.c_int64(n), C.byref(cfg), _p(full), _p(geo), _p(uniq), _p(goff), _p(gidx), _p(counts))
    nf, ng, nu = int(counts[0]), int(counts[1]), int(counts[2])
    groups = [gidx[int(goff[g]):int(goff[g + 1])].copy() for g in range(nu)]
    return {"points_full": full[:nf].copy(), "geometric_idxs": geo[:ng].copy(), "unique_ns": uniq[:nu].copy(),
            "groups": groups, "last_point_ns": int(counts[3])}


POINT_KINDS = ("ouster", "ouster_odyssey", "ouster_r8", "hesai", "livox", "livox_custom2", "velodyne", "velodyne_anybotics", "rslidar")


def point_sizeof(kind: str) -> int:
    L = lib()
    L.ref_point_sizeof.restype = C.c_int64
    return int(L.ref_point_sizeof(POINT_KINDS.index(kind)))


def prepare_input_typed(kind: str, raw, cfg: InputConfig, width=None, height=1, transpose=False, organize=False, header_ts=0.0):
    """Manager::prepareInput<PointT> (src/lidar/manager.cpp:149-383) for any of the reference's point types."""
    L = lib()
    raw = np.ascontiguousarray(raw)
    n = len(raw)
    assert raw.dtype.itemsize == point_sizeof(kind), (raw.dtype.itemsize, point_sizeof(kind))
    width = n if width is None else width
    full = np.zeros(max(n, 1), dtype=np.dtype((np.void, 32)))
    geo = np.zeros(max(n, 1), np.uint64)
    uniq = np.zeros(max(n, 1), np.uint32)
    counts = np.zeros(4, np.uint64)
    L.ref_prepare_input_typed.restype = C.c_int64
    rc = L.ref_prepare_input_typed(POINT_KINDS.index(kind), _p(raw), C.c_int64(n), C.c_uint32(width), C.c_uint32(height), int(transpose),
                                   int(organize), C.c_double(header_ts), C.byref(cfg), _p(full), _p(geo), _p(uniq), _p(counts))
    assert rc >= 0
    nf, ng, nu = int(counts[0]), int(counts[1]), int(counts[2])
    return {"points_full": full[:nf].copy(), "geometric_idxs": geo[:ng].copy(), "unique_ns": uniq[:nu].copy(), "last_point_ns": int(counts[3])}


def downsample(pts, leaf=0.5, max_pts=20, min_dist=0.15):
    pts = np.ascontiguousarray(pts)
    kept = np.empty(len(pts), np.uint32)
    n = lib().ref_downsample(_p(pts), len(pts), leaf, max_pts, min_dist, _p(kept))
    return kept[:n].copy()


def projection_matrix(loc, thresh, evecs):
    P = np.empty((3, 3))
    ax = np.empty(3)
    d = lib().ref_projection_matrix(_p(_f64(loc)), float(thresh), _p(_f64(evecs)), _p(P), _p(ax))
    return bool(d), P, ax
