"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding over the photometric entry points of oracle/libref_cpu.so
(oracle/photo_ref.hpp: the C++ CPU restatement of src/lidar/photometric.cpp, photometric_factor.hpp and
photometric_utils.cpp).  PARITY UNPINNED (the reference has no tests; OpenCV / Eigen / PCL behaviour is restated
from assumptions O1-O12 listed in photo_ref.hpp).  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this.  Same method names as mimosa_amd.capi.Photo / PhotoFactor.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from mimosa_amd.capi import PHOTO_IMAGES, PhotoFeature, PhotoResult, _f64, _p, _PhotoBase, make_photo_config  # struct layouts only

from . import ref_cpu

_SET = False


def lib():
    global _SET
    L = ref_cpu.lib()
    if not _SET:
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.refphoto_create.restype = vp
        L.refphoto_create.argtypes = [vp]
        L.refphoto_destroy.argtypes = [vp]
        L.refphoto_preprocess.argtypes = [vp, vp, vp, sz, vp, vp, sz]
        L.refphoto_preprocess.restype = i32
        L.refphoto_get_image.argtypes = [vp, i32, vp]
        L.refphoto_num_features.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
        L.refphoto_get_features.argtypes = [vp, vp, vp, vp, vp]
        L.refphoto_set_features.argtypes = [vp, vp, sz, vp, vp, vp]
        L.refphoto_detect.argtypes = [vp, i32, vp, vp, vp, sz]
        L.refphoto_update_map.argtypes = [vp, vp, vp, vp, vp, sz]
        L.refphoto_factor_create.restype = vp
        L.refphoto_factor_create.argtypes = [vp, vp, i32]
        L.refphoto_factor_destroy.argtypes = [vp]
        L.refphoto_factor_linearize.argtypes = [vp, vp, vp, vp, vp, C.POINTER(PhotoResult)]
        L.refphoto_factor_get_state.argtypes = [vp, vp, vp, vp]
        _SET = True
    return L


class Photo(_PhotoBase):
    def __init__(self, cfg: dict):
        self.L = lib()
        self.c, self._keep = make_photo_config(cfg)
        self.rows, self.cols = cfg["rows"], cfg["cols"]
        self.h = C.c_void_p(self.L.refphoto_create(C.byref(self.c)))

    def preprocess(self, raw, desk, unique_ns, T_Le_Lt):
        raw, desk = np.ascontiguousarray(raw), np.array(desk, copy=True)
        ns = np.ascontiguousarray(unique_ns, np.uint32)
        T = np.ascontiguousarray(np.asarray(T_Le_Lt, np.float64).reshape(len(ns), 12))
        rc = self.L.refphoto_preprocess(self.h, _p(raw), _p(desk), len(desk), _p(ns), _p(T), len(ns))
        if rc:
            raise RuntimeError("the reference would have thrown in preprocess")
        return desk

    def image(self, name):
        which, dt, k = PHOTO_IMAGES[name]
        out = np.empty((self.rows, self.cols, k) if k > 1 else (self.rows, self.cols), dt)
        self.L.refphoto_get_image(self.h, which, _p(out))
        return out

    def features(self):
        nf, npt = C.c_size_t(), C.c_size_t()
        self.L.refphoto_num_features(self.h, C.byref(nf), C.byref(npt))
        return self._features_from(nf.value, npt.value, lambda f, a, b, c: self.L.refphoto_get_features(self.h, f, a, b, c))

    def set_features(self, features):
        feats, Le, I, psi = self._features_to(features)
        self.L.refphoto_set_features(self.h, feats, len(features), _p(Le), _p(I), _p(psi))

    def detect(self, num, R_W_Be, t_W_Be, bias_directions):
        R, t, b = _f64(R_W_Be), _f64(t_W_Be), np.ascontiguousarray(np.asarray(bias_directions, np.float64).reshape(-1, 3))
        self.L.refphoto_detect(self.h, int(num), _p(R), _p(t), _p(b), len(b))

    def update_map(self, factor, R_W_Be, t_W_Be, bias_directions):
        R, t, b = _f64(R_W_Be), _f64(t_W_Be), np.ascontiguousarray(np.asarray(bias_directions, np.float64).reshape(-1, 3))
        self.L.refphoto_update_map(self.h, factor.h if factor is not None else None, _p(R), _p(t), _p(b), len(b))

    def make_factor(self, VSVt=None, binary=False):
        return PhotoFactor(self, VSVt, binary)

    def __del__(self):
        try:
            self.L.refphoto_destroy(self.h)
        except Exception:
            pass


class PhotoFactor:
    def __init__(self, photo: Photo, VSVt=None, binary=False):
        self.photo, self.L = photo, photo.L
        V = _f64(VSVt) if VSVt is not None else None
        self.h = C.c_void_p(self.L.refphoto_factor_create(photo.h, _p(V), int(binary)))
        self.n = len(photo.features())

    def linearize(self, R_b, t_b, R_a=None, t_a=None) -> dict:
        out = PhotoResult()
        Rb, tb = _f64(R_b), _f64(t_b)
        Ra = _f64(R_a) if R_a is not None else None
        ta = _f64(t_a) if t_a is not None else None
        self.L.refphoto_factor_linearize(self.h, _p(Rb), _p(tb), _p(Ra), _p(ta), C.byref(out))
        return out.as_dict()

    def state(self, rows=True):
        st = np.empty(self.n, np.int32)
        ce = np.empty((self.n, 2))
        rw = np.empty((self.n, 64, 8)) if rows else None
        self.L.refphoto_factor_get_state(self.h, _p(st), _p(ce), _p(rw))
        return st, ce, rw

    def __del__(self):
        try:
            self.L.refphoto_factor_destroy(self.h)
        except Exception:
            pass
