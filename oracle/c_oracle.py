"""numpy/ctypes front end of oracle/msmv_oracle.c.  TEST INFRASTRUCTURE ONLY (same rules as
oracle/sparsebev_oracle.py).  Build with `make -C oracle`."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return _PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def msmv_fwd(feats_cl, loc, weights):
    """feats_cl: list of [B',N,H,W,C] float32 arrays -> [B',Q,C,P]."""
    feats = [_f32(f) for f in feats_cl]
    loc, weights = _f32(loc), _f32(weights)
    Bp, N, _, _, C = feats[0].shape
    _, Q, P, _ = loc.shape
    L = len(feats)
    out = np.empty((Bp, Q, C, P), dtype=np.float32)
    ptrs = (ctypes.c_void_p * L)(*[f.ctypes.data for f in feats])
    hw = np.array([[f.shape[2], f.shape[3]] for f in feats], dtype=np.int32)
    _load().oracle_msmv_fwd(ptrs, hw.ctypes.data_as(ctypes.c_void_p), L, ctypes.c_int64(Bp), N, C, Q, P,
                            loc.ctypes.data_as(ctypes.c_void_p), weights.ctypes.data_as(ctypes.c_void_p),
                            out.ctypes.data_as(ctypes.c_void_p))
    return out


def project(sample_points, lidar2img, image_h, image_w, eps=1e-5, n_views=6):
    """sample_points [B,Q,T,GP,3], lidar2img [B,T*N,4,4] -> uvh [B,T,N,Q,GP,3], valid uint8, iview int32."""
    pts, l2i = _f32(sample_points), _f32(lidar2img)
    B, Q, T, GP, _ = pts.shape
    uvh = np.empty((B, T, n_views, Q, GP, 3), dtype=np.float32)
    valid = np.empty((B, T, n_views, Q, GP), dtype=np.uint8)
    iview = np.empty((B, T, Q, GP), dtype=np.int32)
    _load().oracle_project(pts.ctypes.data_as(ctypes.c_void_p), l2i.ctypes.data_as(ctypes.c_void_p), B, Q, T, n_views, GP,
                           ctypes.c_float(image_h), ctypes.c_float(image_w), ctypes.c_float(eps),
                           uvh.ctypes.data_as(ctypes.c_void_p), valid.ctypes.data_as(ctypes.c_void_p),
                           iview.ctypes.data_as(ctypes.c_void_p))
    return uvh, valid, iview
