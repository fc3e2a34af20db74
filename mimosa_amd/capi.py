"""ctypes binding over libmimosa_hip.so (include/mimosa_hip.h) — test / bench tooling.

Thin and literal: one Python method per C entry point, numpy arrays for the plain buffers.  There is
no fallback of any kind: if the library is missing, or no GPU is present, these raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None

MH_OK, MH_ERR_INVALID_ARG, MH_ERR_HIP, MH_ERR_NO_DEVICE, MH_ERR_OOM, MH_ERR_UNSUPPORTED = range(6)

# every symbol include/mimosa_hip.h declares
EXPORTS = [
    "mh_abi_version", "mh_init", "mh_shutdown", "mh_last_error", "mh_set_profiling",
    "mh_stream", "mh_synchronize", "mh_timer_begin", "mh_timer_end",
    "mh_map_create", "mh_map_insert", "mh_map_insert_device", "mh_map_insert_from_scan", "mh_map_copy", "mh_map_retain", "mh_map_release", "mh_map_get_stats",
    "mh_map_get_cloud", "mh_map_knn",
    "mh_icp_create", "mh_icp_clone", "mh_icp_destroy", "mh_icp_linearize", "mh_icp_linearize_async",
    "mh_icp_wait", "mh_icp_linearize_batch", "mh_icp_get_state", "mh_icp_reset", "mh_icp_set_components", "mh_icp_size",
    "mh_deskew", "mh_transform_f32",
    "mh_scan_create", "mh_scan_destroy", "mh_scan_prepare_input", "mh_scan_prepare_input_device", "mh_scan_prefetch", "mh_scan_prepare_input_prefetched", "mh_scan_prepare_input_layout", "mh_scan_get_unique_ns", "mh_scan_deskew",
    "mh_scan_preprocess_geometric", "mh_scan_get_points", "mh_scan_get_indices", "mh_icp_create_from_scan",
    "mh_init_on_stream", "mh_map_insert_shard", "mh_icp_create_from_device", 
    "mh_shard_unique_id", "mh_shard_comm_init_rccl", "mh_shard_comm_init_local", "mh_shard_comm_destroy", "mh_shard_comm_world", "mh_shard_comm_rank",
    "mh_shard_comm_backend", "mh_shard_comm_info", "mh_shard_owner_of_block", "mh_map_insert_shard_from_scan", "mh_scan_device_points", "mh_shard_icp_create", "mh_shard_icp_linearize", "mh_shard_icp_linearize_async", "mh_shard_icp_linearize_batch",
    "mh_shard_icp_linearize_batch_async", "mh_shard_icp_wait", "mh_shard_icp_reset", "mh_shard_icp_set_components", "mh_shard_icp_get_state",
    "mh_shard_icp_stats", "mh_shard_icp_destroy", "mh_alloc_check_stats",
    "mh_photo_create", "mh_photo_destroy", "mh_photo_preprocess", "mh_scan_keep_raw", "mh_photo_preprocess_scan", "mh_photo_preprocess_scan_begin", "mh_photo_preprocess_commit", "mh_photo_detect_prefetch", "mh_photo_get_image",
    "mh_photo_num_features", "mh_photo_get_features", "mh_photo_set_features", "mh_photo_detect_features", "mh_photo_update_map",
    "mh_photo_factor_create", "mh_photo_factor_clone", "mh_photo_factor_destroy", "mh_photo_factor_linearize", "mh_photo_factor_linearize_async", "mh_photo_factor_wait", "mh_photo_factor_get_state", "mh_photo_factor_size",
]


class MhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mimosa_hip error {code}: {msg}")
        self.code = code


class RegConfig(C.Structure):
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


class MapConfig(C.Structure):
    _fields_ = [
        ("leaf_size", C.c_double), ("min_dist_in_cell", C.c_double), ("max_points_in_cell", C.c_int32),
        ("neighbor_voxel_mode", C.c_int32), ("lru_horizon", C.c_int64), ("lru_clear_cycle", C.c_int32),
        ("reserved", C.c_int32),
    ]


class MapStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_voxels", "n_points", "n_blocks", "device_bytes", "uploads", "upload_bytes",
                                         "delta_uploads", "full_uploads")]


class IcpResult(C.Structure):
    _fields_ = [
        ("H_ss", C.c_double * 36), ("H_st", C.c_double * 36), ("H_tt", C.c_double * 36),
        ("b_s", C.c_double * 6), ("b_t", C.c_double * 6), ("f", C.c_double),
        ("loc_trans_comp", C.c_double * 3), ("loc_rot_comp", C.c_double * 3),
        ("loc_trans_final", C.c_double * 3), ("loc_rot_final", C.c_double * 3),
        ("eigvec_trans", C.c_double * 9), ("eigvec_rot", C.c_double * 9),
        ("degen_rot", C.c_double * 3), ("degen_trans", C.c_double * 3),
        ("degen_eigvec_rot", C.c_double * 9), ("degen_eigvec_trans", C.c_double * 9),
        ("status_hist", C.c_int32 * 9), ("linearize_count", C.c_int32),
        ("mean_candidates", C.c_double), ("mean_scanned", C.c_double), ("n_knn", C.c_int64), ("n_exact_fallback", C.c_int64),
        ("gpu_ms_linearize", C.c_float), ("gpu_ms_localizability", C.c_float),
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


class ShardConfig(C.Structure):
    _fields_ = [("block_log2", C.c_int32), ("force_collectives", C.c_int32)]


class ShardStats(C.Structure):
    _fields_ = [("n_live", C.c_uint64), ("n_slots", C.c_uint64), ("slot_capacity", C.c_uint64), ("n_total", C.c_uint64),
                ("segment_records", C.c_uint32), ("last_max_movers", C.c_uint32), ("retries_total", C.c_uint32), ("retries_last", C.c_uint32),
                ("compactions_total", C.c_uint32), ("collectives_last", C.c_uint32),
                ("world", C.c_int32), ("rank", C.c_int32), ("collective", C.c_int32), ("linearize_count", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


SHARD_UNIQUE_ID_BYTES = 128


class InputConfig(C.Structure):
    """mh_input_config: the ManagerConfig / GeometricConfig fields Manager::prepareInput reads."""
    _fields_ = [
        ("range_min", C.c_float), ("range_max", C.c_float), ("intensity_min", C.c_float), ("intensity_max", C.c_float),
        ("ns_max", C.c_float), ("z_offset", C.c_float), ("create_full_res_pointcloud", C.c_int32),
        ("point_skip_divisor", C.c_int32), ("ring_skip_divisor", C.c_int32),
    ]


class PointLayout(C.Structure):
    """mh_point_layout: where the fields of a sensor's point record sit (what PointCloud2.fields says)."""
    _fields_ = [
        ("stride", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32), ("off_intensity", C.c_uint32),
        ("intensity_is_u16", C.c_int32), ("off_time", C.c_uint32), ("time_kind", C.c_int32), ("off_ring", C.c_uint32),
        ("ring_kind", C.c_int32), ("ring_filter", C.c_int32), ("off_tag", C.c_uint32), ("has_tag", C.c_int32),
    ]


TIME_U32_NS, TIME_F64_S_ABS, TIME_F64_NS_ABS, TIME_F32_S = 0, 1, 2, 3
RING_NONE, RING_U16, RING_U8, RING_F32 = 0, 1, 2, 3


def _dt(stride, **fields):
    names, formats, offsets = zip(*[(k, f, o) for k, (f, o) in fields.items()])
    return np.dtype({"names": list(names), "formats": list(formats), "offsets": list(offsets), "itemsize": stride})


# The reference's point types (include/mimosa/lidar/point.hpp:40-131, EIGEN_ALIGN16 structs): numpy record dtype with the
# members at their C++ offsets, and the mh_point_layout that selects the same branches of Manager::prepareInput<PointT>.
POINT_TYPES = {
    "ouster": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), t=("u4", 20), reflectivity=("u2", 24), ring=("u2", 26)),
               dict(off_intensity=16, off_time=20, time_kind=TIME_U32_NS, off_ring=26, ring_kind=RING_U16, ring_filter=1)),
    "ouster_odyssey": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), t=("u4", 16), reflectivity=("u2", 20), near_ir=("u2", 22)),
                       dict(off_intensity=20, intensity_is_u16=1, off_time=16, time_kind=TIME_U32_NS)),
    "ouster_r8": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), t=("u4", 20), reflectivity=("u2", 24), ring=("u1", 26)),
                  dict(off_intensity=16, off_time=20, time_kind=TIME_U32_NS, off_ring=26, ring_kind=RING_U8, ring_filter=1)),
    "hesai": (_dt(48, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), timestamp=("f8", 24), ring=("u2", 32)),
              dict(off_intensity=16, off_time=24, time_kind=TIME_F64_S_ABS, off_ring=32, ring_kind=RING_U16, ring_filter=1)),
    "livox": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), tag=("u1", 20), line=("u1", 21), timestamp=("f8", 24)),
              dict(off_intensity=16, off_time=24, time_kind=TIME_F64_NS_ABS, off_tag=20, has_tag=1)),
    "livox_custom2": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), t=("u4", 12), intensity=("f4", 16), tag=("u1", 20), line=("u1", 21)),
                      dict(off_intensity=16, off_time=12, time_kind=TIME_U32_NS, off_tag=20, has_tag=1)),
    "velodyne": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), ring=("u2", 20), time=("f4", 24)),
                 dict(off_intensity=16, off_time=24, time_kind=TIME_F32_S, off_ring=20, ring_kind=RING_U16, ring_filter=1)),
    "velodyne_anybotics": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), ring=("f4", 20), time=("f4", 24)),
                           dict(off_intensity=16, off_time=24, time_kind=TIME_F32_S, off_ring=20, ring_kind=RING_F32, ring_filter=0)),
    "rslidar": (_dt(32, x=("f4", 0), y=("f4", 4), z=("f4", 8), intensity=("f4", 16), ring=("u2", 20), timestamp=("f8", 24)),
                dict(off_intensity=16, off_time=24, time_kind=TIME_F64_S_ABS, off_ring=20, ring_kind=RING_U16, ring_filter=1)),
}


def point_dtype(kind: str) -> np.dtype:
    return POINT_TYPES[kind][0]


def point_layout(kind: str) -> PointLayout:
    dt, f = POINT_TYPES[kind]
    return PointLayout(stride=dt.itemsize, off_x=0, off_y=4, off_z=8, **f)


class ScanInfo(C.Structure):
    _fields_ = [
        ("n_in", C.c_uint64), ("n_full", C.c_uint64), ("n_geometric", C.c_uint64), ("n_unique_ns", C.c_uint64),
        ("n_body", C.c_uint64), ("n_downsampled", C.c_uint64), ("last_point_ns", C.c_uint32), ("pad", C.c_uint32),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "pad"}



class PhotoConfig(C.Structure):
    """mh_photo_config (lidar::PhotometricConfig, include/mimosa/lidar/photometric_config.hpp:15-87)."""
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32), ("destagger", C.c_int32),
        ("pixel_shift_by_row", C.c_void_p), ("beam_altitude_angles", C.c_void_p),
        ("range_min", C.c_float), ("range_max", C.c_float),
        ("erosion_buffer", C.c_int32), ("patch_size", C.c_int32), ("margin_size", C.c_int32),
        ("intensity_scale", C.c_float), ("intensity_gamma", C.c_float),
        ("remove_lines", C.c_int32), ("filter_brightness", C.c_int32), ("gaussian_blur", C.c_int32), ("gaussian_blur_size", C.c_int32),
        ("gradient_threshold", C.c_float), ("max_dist_from_mean", C.c_float), ("max_dist_from_plane", C.c_float),
        ("nma_radius", C.c_int32), ("num_features_detect", C.c_int32), ("occlusion_range_diff_threshold", C.c_float),
        ("max_feature_life_time", C.c_int32),
        ("high_pass_fir", C.c_void_p), ("n_high_pass", C.c_int32), ("low_pass_fir", C.c_void_p), ("n_low_pass", C.c_int32),
        ("brightness_window_size", C.c_int32 * 2), ("lidar_origin_to_beam_origin_mm", C.c_float),
        ("rotate_patch_to_align_with_gradient", C.c_int32),
        ("patch_offsets", C.c_void_p), ("n_patch_offsets", C.c_int32),
        ("use_robust_cost_function", C.c_int32), ("robust_cost_function", C.c_int32),
        ("robust_cost_function_parameter", C.c_double), ("error_scale", C.c_double), ("max_error", C.c_double), ("sigma", C.c_double),
        ("T_B_L_R", C.c_double * 9), ("T_B_L_t", C.c_double * 3),
        ("static_mask", C.c_void_p),
    ]


class PhotoFeature(C.Structure):
    _fields_ = [("id", C.c_uint32), ("life_time", C.c_int32), ("n_points", C.c_int32), ("pad", C.c_int32),
                ("center", C.c_double * 2), ("normal", C.c_double * 3), ("mean_intensity", C.c_double), ("sigma_intensity", C.c_double)]


class PhotoResult(C.Structure):
    _fields_ = [
        ("H_bb", C.c_double * 36), ("H_ba", C.c_double * 36), ("H_aa", C.c_double * 36), ("b_b", C.c_double * 6), ("b_a", C.c_double * 6),
        ("f", C.c_double), ("loc_trans_final", C.c_double * 3), ("loc_rot_final", C.c_double * 3),
        ("eigvec_trans", C.c_double * 9), ("eigvec_rot", C.c_double * 9), ("status_hist", C.c_int32 * 9),
        ("n_exceptions", C.c_int32), ("gpu_ms", C.c_float),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = np.array(v) if hasattr(v, "__len__") else v
        for k in ("H_bb", "H_ba", "H_aa"):
            d[k] = d[k].reshape(6, 6)
        for k in ("eigvec_trans", "eigvec_rot"):
            d[k] = d[k].reshape(3, 3)
        return d


PHOTO_IMAGES = {"intensity": (0, np.float32, 1), "range": (1, np.float32, 1), "dx": (2, np.float32, 1), "dy": (3, np.float32, 1),
                "mask": (4, np.uint8, 1), "idx": (5, np.int32, 1), "yaw": (6, np.float32, 1), "proj_idx": (7, np.int32, 10),
                "grad": (8, np.uint8, 1), "detection_mask": (9, np.uint8, 1)}


def make_photo_config(d: dict):
    """Build an mh_photo_config from a dict (mimosa_amd.synth_photo.photo_config()); returns (struct, keep-alive list)."""
    c = PhotoConfig()
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    for k, _ in PhotoConfig._fields_:
        if k in ("pixel_shift_by_row", "patch_offsets"):
            setattr(c, k, arr(d[k], np.int32))
        elif k == "beam_altitude_angles":
            setattr(c, k, arr(d[k], np.float32))
        elif k in ("high_pass_fir", "low_pass_fir"):
            setattr(c, k, arr(d[k], np.float64) if d.get(k) is not None and len(d[k]) else None)
        elif k == "static_mask":
            setattr(c, k, arr(d[k], np.uint8) if d.get(k) is not None else None)
        elif k == "brightness_window_size":
            c.brightness_window_size[0], c.brightness_window_size[1] = int(d[k][0]), int(d[k][1])
        elif k == "T_B_L_R":
            for i, v in enumerate(np.asarray(d[k], float).ravel()):
                c.T_B_L_R[i] = v
        elif k == "T_B_L_t":
            for i, v in enumerate(np.asarray(d[k], float).ravel()):
                c.T_B_L_t[i] = v
        elif k == "n_high_pass":
            c.n_high_pass = len(d["high_pass_fir"]) if d.get("high_pass_fir") is not None else 0
        elif k == "n_low_pass":
            c.n_low_pass = len(d["low_pass_fir"]) if d.get("low_pass_fir") is not None else 0
        elif k == "n_patch_offsets":
            c.n_patch_offsets = len(np.asarray(d["patch_offsets"]).reshape(-1, 2))
        else:
            setattr(c, k, d[k])
    return c, keep


def make_input_config(**kw) -> InputConfig:
    """ENWIDE manager block (config/enwide/params.yaml:66-72) + geometric skip divisors (:79-80) by default."""
    d = dict(range_min=0.2, range_max=100.0, intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9, z_offset=0.0,
             create_full_res_pointcloud=1, point_skip_divisor=4, ring_skip_divisor=1)
    d.update(kw)
    c = InputConfig()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def make_reg_config(**kw) -> RegConfig:
    c = RegConfig()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Load libmimosa_hip.so.  Raises if it cannot be built / found — never substitutes anything."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if not os.path.exists(path):
        if not build_if_missing:
            raise FileNotFoundError(path)
        _build.build()
    # kernel-tuning experiments (tools/variant.sh) load a differently compiled build of the SAME library
    path = os.environ.get("MH_LIB_OVERRIDE", path)
    L = C.CDLL(path)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    pvp = C.POINTER(C.c_void_p)
    L.mh_abi_version.restype = i32
    L.mh_shard_owner_of_block.argtypes = [i32, i32, i32, i32]
    L.mh_shard_owner_of_block.restype = i32
    L.mh_init.argtypes = [i32, pvp]
    L.mh_shutdown.argtypes = [vp]
    L.mh_shutdown.restype = None
    L.mh_last_error.argtypes = [vp]
    L.mh_last_error.restype = C.c_char_p
    L.mh_set_profiling.argtypes = [vp, i32]
    L.mh_stream.argtypes = [vp]
    L.mh_stream.restype = vp
    L.mh_synchronize.argtypes = [vp]
    L.mh_timer_begin.argtypes = [vp]
    L.mh_timer_end.argtypes = [vp, C.POINTER(C.c_float)]
    L.mh_map_create.argtypes = [vp, C.POINTER(MapConfig), pvp]
    L.mh_map_insert.argtypes = [vp, vp, sz, sz]
    L.mh_map_insert_device.argtypes = [vp, vp, sz, sz, vp, vp]
    L.mh_map_insert_from_scan.argtypes = [vp, vp, vp, vp]
    L.mh_map_copy.argtypes = [vp, pvp]
    L.mh_map_retain.argtypes = [vp]
    L.mh_map_release.argtypes = [vp]
    L.mh_map_release.restype = None
    L.mh_map_get_stats.argtypes = [vp, C.POINTER(MapStats)]
    L.mh_map_get_cloud.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.mh_map_knn.argtypes = [vp, vp, sz, i32, vp, vp, vp]
    L.mh_icp_create.argtypes = [vp, vp, vp, sz, C.POINTER(RegConfig), i32, pvp]
    L.mh_icp_clone.argtypes = [vp, pvp]
    L.mh_icp_destroy.argtypes = [vp]
    L.mh_icp_destroy.restype = None
    L.mh_icp_linearize.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(IcpResult)]
    L.mh_icp_linearize_async.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(IcpResult)]
    L.mh_icp_wait.argtypes = [vp]
    L.mh_icp_linearize_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    L.mh_icp_get_state.argtypes = [vp, vp, vp, vp]
    L.mh_icp_reset.argtypes = [vp]
    L.mh_icp_set_components.argtypes = [vp, C.c_int]
    L.mh_icp_size.argtypes = [vp]
    L.mh_icp_size.restype = sz
    L.mh_deskew.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp]
    L.mh_transform_f32.argtypes = [vp, vp, sz, vp, vp]
    L.mh_scan_create.argtypes = [vp, pvp]
    L.mh_scan_destroy.argtypes = [vp]
    L.mh_scan_destroy.restype = None
    L.mh_scan_prepare_input.argtypes = [vp, vp, sz, C.POINTER(InputConfig), C.POINTER(ScanInfo)]
    L.mh_scan_prepare_input_layout.argtypes = [vp, vp, sz, C.POINTER(PointLayout), C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_double,
                                               C.POINTER(InputConfig), C.POINTER(ScanInfo)]
    L.mh_scan_prepare_input_device.argtypes = [vp, vp, sz, C.POINTER(InputConfig), C.POINTER(ScanInfo)]
    L.mh_scan_prefetch.argtypes = [vp, vp, sz]
    L.mh_scan_prepare_input_prefetched.argtypes = [vp, C.POINTER(InputConfig), C.POINTER(ScanInfo)]
    L.mh_scan_get_unique_ns.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.mh_scan_deskew.argtypes = [vp, vp, sz]
    L.mh_scan_preprocess_geometric.argtypes = [vp, vp, vp, C.c_double, i32, C.c_double, C.POINTER(ScanInfo)]
    L.mh_scan_get_points.argtypes = [vp, i32, vp, sz, C.POINTER(sz)]
    L.mh_scan_get_indices.argtypes = [vp, i32, vp, sz, C.POINTER(sz)]
    L.mh_icp_create_from_scan.argtypes = [vp, vp, vp, C.POINTER(RegConfig), i32, pvp]
    L.mh_init_on_stream.argtypes = [i32, vp, pvp]
    L.mh_map_insert_shard.argtypes = [vp, vp, sz, sz, i32, i32, i32]
    L.mh_map_insert_shard_from_scan.argtypes = [vp, vp, vp, vp, i32, i32, i32]
    L.mh_scan_device_points.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz)]
    L.mh_icp_create_from_device.argtypes = [vp, vp, vp, sz, C.POINTER(RegConfig), i32, pvp]
    L.mh_shard_unique_id.argtypes = [vp]
    L.mh_shard_comm_init_rccl.argtypes = [vp, vp, i32, i32, pvp]
    L.mh_shard_comm_init_local.argtypes = [i32, pvp]
    L.mh_shard_comm_destroy.argtypes = [vp]
    L.mh_shard_comm_destroy.restype = None
    L.mh_shard_comm_world.argtypes = [vp]
    L.mh_shard_comm_rank.argtypes = [vp]
    L.mh_shard_comm_backend.argtypes = [vp]
    L.mh_shard_comm_backend.restype = C.c_char_p
    L.mh_shard_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mh_shard_icp_create.argtypes = [vp, vp, vp, vp, sz, i32, C.POINTER(RegConfig), i32, C.POINTER(ShardConfig), pvp]
    L.mh_shard_icp_linearize.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(IcpResult)]
    L.mh_shard_icp_linearize_async.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(IcpResult)]
    L.mh_shard_icp_linearize_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    L.mh_shard_icp_linearize_batch_async.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    L.mh_shard_icp_wait.argtypes = [vp]
    L.mh_shard_icp_reset.argtypes = [vp]
    L.mh_shard_icp_set_components.argtypes = [vp, C.c_int]
    L.mh_shard_icp_get_state.argtypes = [vp, vp, vp, vp, vp, sz, C.POINTER(sz)]
    L.mh_shard_icp_stats.argtypes = [vp, C.POINTER(ShardStats)]
    L.mh_alloc_check_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.mh_shard_icp_destroy.argtypes = [vp]
    L.mh_shard_icp_destroy.restype = None
    L.mh_photo_create.argtypes = [vp, C.POINTER(PhotoConfig), pvp]
    L.mh_photo_destroy.argtypes = [vp]
    L.mh_photo_destroy.restype = None
    L.mh_photo_preprocess.argtypes = [vp, vp, vp, sz, vp, vp, sz]
    L.mh_scan_keep_raw.argtypes = [vp, i32]
    L.mh_photo_preprocess_scan.argtypes = [vp, vp, vp, sz]
    L.mh_photo_preprocess_scan_begin.argtypes = [vp, vp, vp, sz]
    L.mh_photo_preprocess_commit.argtypes = [vp]
    L.mh_photo_detect_prefetch.argtypes = [vp]
    L.mh_photo_get_image.argtypes = [vp, i32, vp, sz]
    L.mh_photo_num_features.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    L.mh_photo_get_features.argtypes = [vp, vp, vp, vp, vp]
    L.mh_photo_set_features.argtypes = [vp, vp, sz, vp, vp, vp]
    L.mh_photo_detect_features.argtypes = [vp, i32, vp, vp, vp, sz]
    L.mh_photo_update_map.argtypes = [vp, vp, vp, vp, vp, sz]
    L.mh_photo_factor_create.argtypes = [vp, vp, i32, pvp]
    L.mh_photo_factor_clone.argtypes = [vp, pvp]
    L.mh_photo_factor_linearize_async.argtypes = [vp, vp, vp, vp, vp]
    L.mh_photo_factor_wait.argtypes = [vp, vp]
    L.mh_photo_factor_destroy.argtypes = [vp]
    L.mh_photo_factor_destroy.restype = None
    L.mh_photo_factor_linearize.argtypes = [vp, vp, vp, vp, vp, C.POINTER(PhotoResult)]
    L.mh_photo_factor_get_state.argtypes = [vp, vp, vp, vp]
    L.mh_photo_factor_size.argtypes = [vp]
    L.mh_photo_factor_size.restype = sz
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.mh_init(device, C.byref(h))
        if rc != MH_OK:
            raise MhError(rc, (self.L.mh_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self._children = 0
        self._closing = False

    def _child_released(self):
        self._children -= 1
        if self._closing and self._children == 0:
            self.close()

    def check(self, rc):
        if rc != MH_OK:
            raise MhError(rc, (self.L.mh_last_error(self.h) or b"").decode())

    def set_profiling(self, every):
        """0/False = off, n = HIP events around the kernels of every n-th linearize call (True = every call)."""
        self.check(self.L.mh_set_profiling(self.h, int(every)))

    def synchronize(self):
        self.check(self.L.mh_synchronize(self.h))

    def timer_begin(self):
        self.check(self.L.mh_timer_begin(self.h))

    def timer_end(self) -> float:
        ms = C.c_float()
        self.check(self.L.mh_timer_end(self.h, C.byref(ms)))
        return ms.value

    def deskew(self, pts, unique_ns, Rt12, R_B_L=None, t_B_L=None):
        pts = np.ascontiguousarray(pts).copy()
        u = np.ascontiguousarray(unique_ns, dtype=np.uint32)
        P = np.ascontiguousarray(Rt12, dtype=np.float32)
        Rb = np.ascontiguousarray(R_B_L, dtype=np.float32) if R_B_L is not None else None
        tb = np.ascontiguousarray(t_B_L, dtype=np.float32) if t_B_L is not None else None
        self.check(self.L.mh_deskew(self.h, _p(pts), len(pts), _p(u), _p(P), len(u), _p(Rb), _p(tb)))
        return pts

    def transform_f32(self, pts, R, t):
        pts = np.ascontiguousarray(pts).copy()
        R = np.ascontiguousarray(R, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        self.check(self.L.mh_transform_f32(self.h, _p(pts), len(pts), _p(R), _p(t)))
        return pts

    def close(self):
        """Shut the context down; deferred until every map / factor created on it is released."""
        if getattr(self, "h", None):
            if self._children > 0:
                self._closing = True
                return
            self.L.mh_shutdown(self.h)
            self.h = None


class VoxelMap:
    """IncrementalVoxelMapPCL counterpart (include/mimosa/lidar/incremental_voxel_map.hpp:22-54)."""

    def __init__(self, ctx: Context, leaf=0.5, min_dist=0.15, max_pts=20, mode=19, lru_horizon=1000,
                 lru_clear_cycle=10, _h=None):
        self.ctx, self.L = ctx, ctx.L
        ctx._children += 1
        if _h is not None:
            self.h = _h
            return
        cfg = MapConfig(leaf, min_dist, max_pts, mode, lru_horizon, lru_clear_cycle, 0)
        h = C.c_void_p()
        ctx.check(self.L.mh_map_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def insert(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.ctx.check(self.L.mh_map_insert(self.h, _p(xyz), xyz.shape[0], 3))

    def insert_from_scan(self, scan, R_W_Be, t_W_Be):
        R = np.ascontiguousarray(R_W_Be, np.float32)
        t = np.ascontiguousarray(t_W_Be, np.float32)
        self.ctx.check(self.L.mh_map_insert_from_scan(self.h, scan.h, _p(R), _p(t)))

    def copy(self):
        h = C.c_void_p()
        self.ctx.check(self.L.mh_map_copy(self.h, C.byref(h)))
        return VoxelMap(self.ctx, _h=h)

    def stats(self) -> dict:
        s = MapStats()
        self.ctx.check(self.L.mh_map_get_stats(self.h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in s._fields_}

    def get_cloud(self):
        n = C.c_size_t()
        self.ctx.check(self.L.mh_map_get_cloud(self.h, None, 0, C.byref(n)))
        xyz = np.empty((n.value, 3), np.float32)
        self.ctx.check(self.L.mh_map_get_cloud(self.h, _p(xyz), n.value, C.byref(n)))
        return xyz

    def knn(self, q, k=5):
        q = _f64(q).reshape(-1, 3)
        n = q.shape[0]
        pts = np.empty((n, k, 3))
        sq = np.empty((n, k))
        found = np.empty(n, np.int32)
        self.ctx.check(self.L.mh_map_knn(self.h, _p(q), n, k, _p(pts), _p(sq), _p(found)))
        return pts, sq, found

    def release(self):
        if getattr(self, "h", None):
            self.L.mh_map_release(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Scan:
    """Device-resident scan front end: Manager::prepareInput -> deskewPoints -> Geometric::preprocess
    (src/lidar/manager.cpp:149-512, src/lidar/geometric.cpp:55-183) with the cloud staying on the GPU."""
    FULL, BODY, DOWNSAMPLED = 0, 1, 2

    def __init__(self, ctx: Context):
        self.ctx, self.L = ctx, ctx.L
        ctx._children += 1
        h = C.c_void_p()
        ctx.check(self.L.mh_scan_create(ctx.h, C.byref(h)))
        self.h = h

    def prepare_input(self, raw, cfg: InputConfig) -> dict:
        raw = np.ascontiguousarray(raw)
        assert raw.dtype.itemsize == 32
        info = ScanInfo()
        self.ctx.check(self.L.mh_scan_prepare_input(self.h, _p(raw), len(raw), C.byref(cfg), C.byref(info)))
        return info.as_dict()

    def prepare_input_layout(self, raw, layout: PointLayout, cfg: InputConfig, width=None, height=1, transpose=False, organize_by_ring=False,
                             header_ts=0.0) -> dict:
        """Manager::prepareInput<PointT> for any point record described by `layout` (see POINT_TYPES)."""
        raw = np.ascontiguousarray(raw)
        assert raw.dtype.itemsize == layout.stride
        n = len(raw)
        width = n if width is None else width
        info = ScanInfo()
        self.ctx.check(self.L.mh_scan_prepare_input_layout(self.h, _p(raw), n, C.byref(layout), width, height, int(transpose),
                                                           int(organize_by_ring), header_ts, C.byref(cfg), C.byref(info)))
        return info.as_dict()

    def prefetch(self, raw):
        """stage the next cloud: pinned copy + host-to-device copy on the handle's copy stream (returns before the copy is done)"""
        raw = np.ascontiguousarray(raw)
        assert raw.dtype.itemsize == 32
        self.ctx.check(self.L.mh_scan_prefetch(self.h, _p(raw), len(raw)))

    def prepare_input_prefetched(self, cfg: InputConfig) -> dict:
        info = ScanInfo()
        self.ctx.check(self.L.mh_scan_prepare_input_prefetched(self.h, C.byref(cfg), C.byref(info)))
        return info.as_dict()

    def prepare_input_device(self, d_raw_ptr: int, n: int, cfg: InputConfig) -> dict:
        """raw cloud (32-byte Ouster records) already in device memory at address d_raw_ptr"""
        info = ScanInfo()
        self.ctx.check(self.L.mh_scan_prepare_input_device(self.h, C.c_void_p(d_raw_ptr), n, C.byref(cfg), C.byref(info)))
        return info.as_dict()

    def unique_ns(self):
        n = C.c_size_t()
        self.ctx.check(self.L.mh_scan_get_unique_ns(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32)
        self.ctx.check(self.L.mh_scan_get_unique_ns(self.h, _p(out), out.size, C.byref(n)))
        return out

    def deskew(self, Rt12):
        P = np.ascontiguousarray(Rt12, dtype=np.float32).reshape(-1, 12)
        self.ctx.check(self.L.mh_scan_deskew(self.h, _p(P), len(P)))

    def preprocess_geometric(self, R_B_L, t_B_L, leaf=0.5, max_pts=20, min_dist=0.15) -> dict:
        R = np.ascontiguousarray(R_B_L, dtype=np.float32)
        t = np.ascontiguousarray(t_B_L, dtype=np.float32)
        info = ScanInfo()
        self.ctx.check(self.L.mh_scan_preprocess_geometric(self.h, _p(R), _p(t), float(leaf), int(max_pts), float(min_dist),
                                                           C.byref(info)))
        return info.as_dict()

    def points(self, which):
        from .synth import POINT_DTYPE
        n = C.c_size_t()
        self.ctx.check(self.L.mh_scan_get_points(self.h, which, None, 0, C.byref(n)))
        out = np.zeros(n.value, POINT_DTYPE)
        self.ctx.check(self.L.mh_scan_get_points(self.h, which, _p(out), out.size, C.byref(n)))
        return out

    def indices(self, which):
        n = C.c_size_t()
        self.ctx.check(self.L.mh_scan_get_indices(self.h, which, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32)
        self.ctx.check(self.L.mh_scan_get_indices(self.h, which, _p(out), out.size, C.byref(n)))
        return out

    def make_factor(self, map_, cfg: RegConfig, binary=False):
        h = C.c_void_p()
        self.ctx.check(self.L.mh_icp_create_from_scan(self.ctx.h, map_.h, self.h, C.byref(cfg), int(binary), C.byref(h)))
        n = int(self.L.mh_icp_size(h))
        f = ICPFactor(self.ctx, map_, None, None, _h=h, _n=n)
        f._keep = []
        return f

    def destroy(self):
        if getattr(self, "h", None):
            self.L.mh_scan_destroy(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def linearize_batch(factors, Rs, ts, g_units=None, R_tgts=None, t_tgts=None) -> list:
    """mh_icp_linearize_batch: every factor of `factors` linearized at its own pose in one K3 + one K4 launch."""
    n = len(factors)
    ctx = factors[0].ctx
    R = np.ascontiguousarray(np.asarray(Rs, np.float64).reshape(n, 9))
    t = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(n, 3))
    g = np.ascontiguousarray(np.tile(np.array([0.0, 0.0, -1.0]), (n, 1)) if g_units is None else np.asarray(g_units, np.float64).reshape(n, 3))
    Rt = None if R_tgts is None else np.ascontiguousarray(np.asarray(R_tgts, np.float64).reshape(n, 9))
    tt = None if t_tgts is None else np.ascontiguousarray(np.asarray(t_tgts, np.float64).reshape(n, 3))
    handles = (C.c_void_p * n)(*[f.h for f in factors])
    out = (IcpResult * n)()
    ctx.check(ctx.L.mh_icp_linearize_batch(handles, n, _p(R), _p(t), _p(Rt), _p(tt), _p(g), out))
    return [out[i].as_dict() for i in range(n)]


class ICPFactor:
    """lidar::ICPFactor counterpart (include/mimosa/lidar/geometric_factor.hpp:25-563)."""

    def __init__(self, ctx: Context, map_: VoxelMap, pts, cfg: RegConfig, binary=False, _h=None, _n=None):
        self.ctx, self.L, self.map = ctx, ctx.L, map_
        ctx._children += 1
        if _h is not None:
            self.h, self.n = _h, _n
            return
        pts = np.ascontiguousarray(pts)
        assert pts.dtype.itemsize == 32
        h = C.c_void_p()
        ctx.check(self.L.mh_icp_create(ctx.h, map_.h, _p(pts), len(pts), C.byref(cfg), int(binary), C.byref(h)))
        self.h, self.n = h, len(pts)
        self._keep = []

    def clone(self):
        h = C.c_void_p()
        self.ctx.check(self.L.mh_icp_clone(self.h, C.byref(h)))
        c = ICPFactor(self.ctx, self.map, None, None, _h=h, _n=self.n)
        c._keep = []
        return c

    def linearize(self, R, t, g_unit=(0.0, 0.0, -1.0), R_tgt=None, t_tgt=None) -> dict:
        out = IcpResult()
        R, t, g = _f64(R), _f64(t), _f64(g_unit)
        Rt = _f64(R_tgt) if R_tgt is not None else None
        tt = _f64(t_tgt) if t_tgt is not None else None
        self.ctx.check(self.L.mh_icp_linearize(self.h, _p(R), _p(t), _p(Rt), _p(tt), _p(g), C.byref(out)))
        return out.as_dict()

    def linearize_async(self, R, t, g_unit=(0.0, 0.0, -1.0)) -> IcpResult:
        out = IcpResult()
        R, t, g = _f64(R), _f64(t), _f64(g_unit)
        self._keep.append((out, R, t, g))
        self.ctx.check(self.L.mh_icp_linearize_async(self.h, _p(R), _p(t), None, None, _p(g), C.byref(out)))
        return out

    def wait(self):
        self.ctx.check(self.L.mh_icp_wait(self.h))
        self._keep = []

    def reset(self):
        self.ctx.check(self.L.mh_icp_reset(self.h))

    def set_components(self, enabled: bool):
        """component localizabilities + status histogram (K4) on / off for the following linearize calls"""
        self.ctx.check(self.L.mh_icp_set_components(self.h, int(bool(enabled))))

    def state(self):
        st = np.empty(self.n, np.int32)
        means = np.empty((self.n, 3))
        normals = np.empty((self.n, 3))
        self.ctx.check(self.L.mh_icp_get_state(self.h, _p(st), _p(means), _p(normals)))
        return st, means, normals

    def destroy(self):
        if getattr(self, "h", None):
            self.L.mh_icp_destroy(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ShardComm:
    """mh_shard_comm: the communicator of the native map-sharded factor.  rccl(): one process per GPU, the 128-byte
    ncclUniqueId travels through whatever channel the caller has (torch.distributed in bench.py); local(): `world` ranks inside
    this process, one host thread each (tests on a one-GPU box)."""

    def __init__(self, L, h):
        self.L, self.h = L, h

    @staticmethod
    def unique_id() -> bytes:
        L = load()
        buf = C.create_string_buffer(SHARD_UNIQUE_ID_BYTES)
        rc = L.mh_shard_unique_id(buf)
        if rc != MH_OK:
            raise MhError(rc, (L.mh_last_error(None) or b"").decode())
        return buf.raw

    @staticmethod
    def rccl(ctx: "Context", unique_id: bytes, world: int, rank: int) -> "ShardComm":
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), SHARD_UNIQUE_ID_BYTES)
        ctx.check(ctx.L.mh_shard_comm_init_rccl(ctx.h, buf, world, rank, C.byref(h)))
        return ShardComm(ctx.L, h)

    @staticmethod
    def local(world: int) -> list:
        L = load()
        arr = (C.c_void_p * world)()
        rc = L.mh_shard_comm_init_local(world, arr)
        if rc != MH_OK:
            raise MhError(rc, (L.mh_last_error(None) or b"").decode())
        return [ShardComm(L, C.c_void_p(arr[r])) for r in range(world)]

    @property
    def world(self):
        return int(self.L.mh_shard_comm_world(self.h))

    @property
    def rank(self):
        return int(self.L.mh_shard_comm_rank(self.h))

    @property
    def backend(self):
        return self.L.mh_shard_comm_backend(self.h).decode()

    def info(self) -> dict:
        n, v = C.c_int(0), C.c_int(0)
        self.L.mh_shard_comm_info(self.h, C.byref(n), C.byref(v))
        return {"ranks_in_communicator": n.value, "rccl_version": v.value}

    def destroy(self):
        if self.h:
            self.L.mh_shard_comm_destroy(self.h)
            self.h = None


class ShardedICPFactor:
    """mh_shard_icp: this rank's part of a scan-to-map factor whose map is sharded across GPUs; linearize() is collective
    and returns the GLOBAL result on every rank."""

    def __init__(self, ctx: Context, comm: ShardComm, shard_map: VoxelMap, pts, cfg: RegConfig, binary=False, block_log2=3, force_collectives=False,
                 d_points_ptr=None, n_device=0):
        self.ctx, self.L, self.comm, self.map = ctx, ctx.L, comm, shard_map
        ctx._children += 1
        sc = ShardConfig(block_log2, int(bool(force_collectives)))
        h = C.c_void_p()
        if d_points_ptr is not None:
            ctx.check(self.L.mh_shard_icp_create(ctx.h, comm.h, shard_map.h, C.c_void_p(d_points_ptr), n_device, 1, C.byref(cfg), int(binary), C.byref(sc), C.byref(h)))
        else:
            pts = np.ascontiguousarray(pts)
            assert pts.dtype.itemsize == 32
            ctx.check(self.L.mh_shard_icp_create(ctx.h, comm.h, shard_map.h, _p(pts), len(pts), 0, C.byref(cfg), int(binary), C.byref(sc), C.byref(h)))
        self.h = h

    def linearize(self, R, t, g_unit=(0.0, 0.0, -1.0), R_tgt=None, t_tgt=None) -> dict:
        out = IcpResult()
        R, t, g = _f64(R), _f64(t), _f64(g_unit)
        Rt = _f64(R_tgt) if R_tgt is not None else None
        tt = _f64(t_tgt) if t_tgt is not None else None
        self.ctx.check(self.L.mh_shard_icp_linearize(self.h, _p(R), _p(t), _p(Rt), _p(tt), _p(g), C.byref(out)))
        return out.as_dict()

    def linearize_async(self, R, t, g_unit=(0.0, 0.0, -1.0), R_tgt=None, t_tgt=None) -> IcpResult:
        """mh_shard_icp_linearize_async: enqueue only; the returned IcpResult is filled by wait() (collective, like linearize)."""
        out = IcpResult()
        R, t, g = _f64(R), _f64(t), _f64(g_unit)
        Rt = _f64(R_tgt) if R_tgt is not None else None
        tt = _f64(t_tgt) if t_tgt is not None else None
        self.ctx.check(self.L.mh_shard_icp_linearize_async(self.h, _p(R), _p(t), _p(Rt), _p(tt), _p(g), C.byref(out)))
        self._pending = getattr(self, "_pending", [])
        self._pending.append(out)  # the library writes into it at wait()
        return out

    def wait(self):
        """mh_shard_icp_wait: completes every round in flight on this factor's communicator."""
        self.ctx.check(self.L.mh_shard_icp_wait(self.h))
        self._pending = []

    def reset(self):
        self.ctx.check(self.L.mh_shard_icp_reset(self.h))

    def set_components(self, enabled: bool):
        self.ctx.check(self.L.mh_shard_icp_set_components(self.h, int(bool(enabled))))

    def stats(self) -> dict:
        s = ShardStats()
        self.ctx.check(self.L.mh_shard_icp_stats(self.h, C.byref(s)))
        return s.as_dict()

    def state(self):
        """(origin, status, mean, normal) of the points this rank holds now."""
        n = C.c_size_t()
        self.ctx.check(self.L.mh_shard_icp_get_state(self.h, None, None, None, None, 0, C.byref(n)))
        n = int(n.value)
        origin, st = np.empty(n, np.uint64), np.empty(n, np.int32)
        mean, nrm = np.empty((n, 3)), np.empty((n, 3))
        m = C.c_size_t()
        self.ctx.check(self.L.mh_shard_icp_get_state(self.h, _p(origin), _p(st), _p(mean), _p(nrm), n, C.byref(m)))
        assert int(m.value) == n
        return origin, st, mean, nrm

    def destroy(self):
        if getattr(self, "h", None):
            self.L.mh_shard_icp_destroy(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class _ShardBatchArgs:
    """Marshalled arguments of a sharded batch call (kept alive until the results are in)."""

    def __init__(self, factors, Rs, ts, g_units, R_tgts, t_tgts):
        n = len(factors)
        self.n = n
        self.R = np.ascontiguousarray(np.asarray(Rs, np.float64).reshape(n, 9))
        self.t = np.ascontiguousarray(np.asarray(ts, np.float64).reshape(n, 3))
        self.g = np.ascontiguousarray(np.tile(np.array([0.0, 0.0, -1.0]), (n, 1)) if g_units is None else np.asarray(g_units, np.float64).reshape(n, 3))
        self.Rt = None if R_tgts is None else np.ascontiguousarray(np.asarray(R_tgts, np.float64).reshape(n, 9))
        self.tt = None if t_tgts is None else np.ascontiguousarray(np.asarray(t_tgts, np.float64).reshape(n, 3))
        self.handles = (C.c_void_p * n)(*[f.h for f in factors])
        self.out = (IcpResult * n)()

    def args(self):
        return (self.handles, self.n, _p(self.R), _p(self.t), _p(self.Rt), _p(self.tt), _p(self.g), self.out)

    def results(self) -> list:
        return [self.out[i].as_dict() for i in range(self.n)]


def sharded_linearize_batch(factors, Rs, ts, g_units=None, R_tgts=None, t_tgts=None) -> list:
    """mh_shard_icp_linearize_batch: the sharded factors of `factors` (one communicator) in ONE protocol round; blocks; collective."""
    a = _ShardBatchArgs(factors, Rs, ts, g_units, R_tgts, t_tgts)
    factors[0].ctx.check(factors[0].L.mh_shard_icp_linearize_batch(*a.args()))
    return a.results()


def sharded_linearize_batch_async(factors, Rs, ts, g_units=None, R_tgts=None, t_tgts=None) -> _ShardBatchArgs:
    """mh_shard_icp_linearize_batch_async: enqueue one round; `.results()` of the returned object is valid after factors[0].wait()."""
    a = _ShardBatchArgs(factors, Rs, ts, g_units, R_tgts, t_tgts)
    factors[0].ctx.check(factors[0].L.mh_shard_icp_linearize_batch_async(*a.args()))
    return a


def alloc_check_stats():
    """(blocks verified at hand-out, words found overwritten) of the allocation cache's MH_ALLOC_CHECK mode; None when it is off."""
    a, b = C.c_ulonglong(), C.c_ulonglong()
    if load().mh_alloc_check_stats(C.byref(a), C.byref(b)) != 0:
        return None
    return int(a.value), int(b.value)


def map_insert_shard_from_scan(ctx: Context, vmap: VoxelMap, scan, R_W_Be, t_W_Be, world: int, rank: int, block_log2: int = 3):
    """mh_map_insert_shard_from_scan: one rank's share of Geometric::updateMap's insert of the resident scan's Be_cloud_, on the device."""
    R = np.ascontiguousarray(R_W_Be, np.float32)
    t = np.ascontiguousarray(t_W_Be, np.float32)
    ctx.check(ctx.L.mh_map_insert_shard_from_scan(vmap.h, scan.h, _p(R), _p(t), world, rank, block_log2))


def map_insert_shard(ctx: Context, vmap: VoxelMap, xyz, world: int, rank: int, block_log2: int = 3):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    ctx.check(ctx.L.mh_map_insert_shard(vmap.h, _p(xyz), len(xyz), 3, world, rank, block_log2))


class _PhotoBase:
    """Shared marshalling of the photometric entry points: the product (libmimosa_hip.so, prefix mh_photo) and the
    checker (oracle/photo_ref.py, prefix refphoto) export the same shapes."""

    def _features_from(self, nf, npt, get):
        feats = (PhotoFeature * max(nf, 1))()
        Le, I, psi = np.zeros((max(npt, 1), 3)), np.zeros(max(npt, 1)), np.zeros(max(npt, 1))
        get(feats, _p(Le), _p(I), _p(psi))
        out, o = [], 0
        for i in range(nf):
            m = feats[i].n_points
            out.append(dict(id=int(feats[i].id), life_time=int(feats[i].life_time), center=np.array(feats[i].center),
                            normal=np.array(feats[i].normal), mean_intensity=feats[i].mean_intensity,
                            sigma_intensity=feats[i].sigma_intensity, Le_ps=Le[o:o + m].copy(), intensities=I[o:o + m].copy(),
                            psi=psi[o:o + m].copy()))
            o += m
        return out

    @staticmethod
    def _features_to(features):
        nf = len(features)
        feats = (PhotoFeature * max(nf, 1))()
        Le, I, psi = [], [], []
        for i, f in enumerate(features):
            feats[i].id, feats[i].life_time, feats[i].n_points = f["id"], f["life_time"], len(f["Le_ps"])
            feats[i].center[0], feats[i].center[1] = f["center"]
            for k in range(3):
                feats[i].normal[k] = f["normal"][k]
            feats[i].mean_intensity, feats[i].sigma_intensity = f["mean_intensity"], f["sigma_intensity"]
            Le.append(np.asarray(f["Le_ps"], float).reshape(-1, 3))
            I.append(np.asarray(f["intensities"], float))
            psi.append(np.asarray(f["psi"], float))
        Le = np.ascontiguousarray(np.concatenate(Le)) if nf else np.zeros((1, 3))
        I = np.ascontiguousarray(np.concatenate(I)) if nf else np.zeros(1)
        psi = np.ascontiguousarray(np.concatenate(psi)) if nf else np.zeros(1)
        return feats, Le, I, psi


class Photo(_PhotoBase):
    """lidar::Photometric counterpart (include/mimosa/lidar/photometric.hpp:22-86) over mh_photo_*."""

    def __init__(self, ctx: Context, cfg: dict):
        self.ctx, self.L = ctx, ctx.L
        self.cfgd = cfg
        self.c, self._keep = make_photo_config(cfg)
        self.rows, self.cols = cfg["rows"], cfg["cols"]
        h = C.c_void_p()
        ctx.check(self.L.mh_photo_create(ctx.h, C.byref(self.c), C.byref(h)))
        self.h = h
        ctx._children += 1

    def preprocess(self, raw, desk, unique_ns, T_Le_Lt):
        """Returns the deskewed cloud with the corrected intensities (a copy)."""
        raw, desk = np.ascontiguousarray(raw), np.array(desk, copy=True)
        ns = np.ascontiguousarray(unique_ns, np.uint32)
        T = np.ascontiguousarray(np.asarray(T_Le_Lt, np.float64).reshape(len(ns), 12))
        self.ctx.check(self.L.mh_photo_preprocess(self.h, _p(raw), _p(desk), len(desk), _p(ns), _p(T), len(ns)))
        return desk

    def preprocess_scan(self, scan, T_Le_Lt):
        T = np.ascontiguousarray(np.asarray(T_Le_Lt, np.float64).reshape(-1, 12))
        self.ctx.check(self.L.mh_photo_preprocess_scan(self.h, scan.h, _p(T), len(T)))

    def preprocess_scan_begin(self, scan, T_Le_Lt):
        """Builds the frame without making it current (may overlap update_map of the previous scan on another thread)."""
        T = np.ascontiguousarray(np.asarray(T_Le_Lt, np.float64).reshape(-1, 12))
        self.ctx.check(self.L.mh_photo_preprocess_scan_begin(self.h, scan.h, _p(T), len(T)))

    def preprocess_commit(self):
        self.ctx.check(self.L.mh_photo_preprocess_commit(self.h))

    def detect_prefetch(self):
        self.ctx.check(self.L.mh_photo_detect_prefetch(self.h))

    def image(self, name):
        which, dt, k = PHOTO_IMAGES[name]
        out = np.empty((self.rows, self.cols, k) if k > 1 else (self.rows, self.cols), dt)
        self.ctx.check(self.L.mh_photo_get_image(self.h, which, _p(out), out.nbytes))
        return out

    def features(self):
        nf, npt = C.c_size_t(), C.c_size_t()
        self.ctx.check(self.L.mh_photo_num_features(self.h, C.byref(nf), C.byref(npt)))
        return self._features_from(nf.value, npt.value,
                                   lambda f, a, b, c: self.ctx.check(self.L.mh_photo_get_features(self.h, f, a, b, c)))

    def set_features(self, features):
        feats, Le, I, psi = self._features_to(features)
        self.ctx.check(self.L.mh_photo_set_features(self.h, feats, len(features), _p(Le), _p(I), _p(psi)))

    def detect(self, num, R_W_Be, t_W_Be, bias_directions):
        R, t, b = _f64(R_W_Be), _f64(t_W_Be), np.ascontiguousarray(np.asarray(bias_directions, np.float64).reshape(-1, 3))
        self.ctx.check(self.L.mh_photo_detect_features(self.h, int(num), _p(R), _p(t), _p(b), len(b)))

    def update_map(self, factor, R_W_Be, t_W_Be, bias_directions):
        R, t, b = _f64(R_W_Be), _f64(t_W_Be), np.ascontiguousarray(np.asarray(bias_directions, np.float64).reshape(-1, 3))
        self.ctx.check(self.L.mh_photo_update_map(self.h, factor.h if factor is not None else None, _p(R), _p(t), _p(b), len(b)))

    def make_factor(self, VSVt=None, binary=False):
        return PhotoFactor(self, VSVt, binary)

    def destroy(self):
        if getattr(self, "h", None):
            self.L.mh_photo_destroy(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class PhotoFactor:
    """lidar::PhotometricFactor counterpart (include/mimosa/lidar/photometric_factor.hpp:22-357)."""

    def __init__(self, photo: Photo, VSVt=None, binary=False, _clone_of=None):
        self.photo, self.ctx, self.L = photo, photo.ctx, photo.L
        V = _f64(VSVt) if VSVt is not None else None
        h = C.c_void_p()
        if _clone_of is not None:
            self.ctx.check(self.L.mh_photo_factor_clone(_clone_of.h, C.byref(h)))
        else:
            self.ctx.check(self.L.mh_photo_factor_create(photo.h, _p(V), int(binary), C.byref(h)))
        self.h = h
        self.n = int(self.L.mh_photo_factor_size(h))
        self.ctx._children += 1

    def linearize(self, R_b, t_b, R_a=None, t_a=None) -> dict:
        out = PhotoResult()
        Rb, tb = _f64(R_b), _f64(t_b)
        Ra = _f64(R_a) if R_a is not None else None
        ta = _f64(t_a) if t_a is not None else None
        self.ctx.check(self.L.mh_photo_factor_linearize(self.h, _p(Rb), _p(tb), _p(Ra), _p(ta), C.byref(out)))
        return out.as_dict()

    def linearize_async(self, R_b, t_b, R_a=None, t_a=None):
        Rb, tb = _f64(R_b), _f64(t_b)
        Ra = _f64(R_a) if R_a is not None else None
        ta = _f64(t_a) if t_a is not None else None
        self.ctx.check(self.L.mh_photo_factor_linearize_async(self.h, _p(Rb), _p(tb), _p(Ra), _p(ta)))

    def wait(self) -> dict:
        out = PhotoResult()
        self.ctx.check(self.L.mh_photo_factor_wait(self.h, C.byref(out)))
        return out.as_dict()

    def clone(self) -> "PhotoFactor":
        """clone() (photometric_factor.hpp:120-124)"""
        return PhotoFactor(self.photo, _clone_of=self)

    def state(self, rows=True):
        st = np.empty(self.n, np.int32)
        ce = np.empty((self.n, 2))
        rw = np.empty((self.n, 64, 8)) if rows else None
        self.ctx.check(self.L.mh_photo_factor_get_state(self.h, _p(st), _p(ce), _p(rw)))
        return st, ce, rw

    def destroy(self):
        if getattr(self, "h", None):
            self.L.mh_photo_factor_destroy(self.h)
            self.h = None
            self.ctx._child_released()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
