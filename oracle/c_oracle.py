"""ctypes front-end of oracle/mvx_oracle.c (TEST INFRASTRUCTURE ONLY; see that file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmvx_oracle.so")
_lib = None


class _Params(C.Structure):
    _fields_ = [
        ("resolution", C.c_double),
        ("dimension", C.c_int32),
        ("blockdim", C.c_int32),
        ("density", C.c_int32),
        ("radii_mode", C.c_int32),
        ("sigma", C.c_double),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mvx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmvx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        pp = C.POINTER(_Params)
        L.ovx_forward_features.argtypes = [pp, dp, fp, C.c_double, fp, C.c_int64, C.c_int32, fp]
        L.ovx_forward_types.argtypes = [pp, dp, ip, C.c_double, fp, C.c_int64, C.c_int32, fp]
        L.ovx_forward_single.argtypes = [pp, dp, C.c_double, fp, C.c_int64, fp]
        L.ovx_num_threads.restype = C.c_int
        L.ovx_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def num_threads() -> int:
    return lib().ovx_num_threads()


def set_num_threads(n: int) -> None:
    lib().ovx_set_num_threads(n)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def voxelize(
    coords,
    channels,
    radii,
    *,
    resolution=0.5,
    dimension=64,
    blockdim=None,
    radii_type="scalar",
    density="gaussian",
    sigma=0.5,
    num_channels=None,
    out=None,
):
    """Same call shape as numpy_port.voxelize: coords after centring/transform;
    channels None | int (V,) | float (V, C); radii python float | (V,) | (C,)."""
    L = lib()
    D = dimension
    xyz = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 3)
    N = xyz.shape[0]
    mode = "single" if channels is None else ("types" if np.ndim(channels) == 1 else "features")
    r_scalar, rad = 0.0, None
    if np.isscalar(radii):
        r_scalar, rmode = float(radii), 0
    else:
        rad = np.ascontiguousarray(radii, dtype=np.float32)
        rmode = 1
    if mode == "features":
        feat = np.ascontiguousarray(channels, dtype=np.float32)
        nC = feat.shape[1]
        if radii_type == "channel-wise":
            rmode = 2
    elif mode == "types":
        types = np.ascontiguousarray(np.asarray(channels).astype(np.int16), dtype=np.int32)
        if num_channels is not None:
            nC = num_channels
        elif radii_type == "channel-wise":
            nC = rad.shape[0]
        else:
            nC = int(types.max()) + 1
        if radii_type == "channel-wise":
            rad = np.ascontiguousarray(rad[types])
    else:
        nC = 1
    P = _Params(resolution, D, 8 if blockdim is None else blockdim, 1 if density == "binary" else 0, rmode, sigma)
    if out is None:
        out = np.empty((nC, D, D, D), dtype=np.float32)
    assert out.flags.c_contiguous and out.dtype == np.float32 and out.shape == (nC, D, D, D)
    po = _ptr(out, C.c_float)
    if mode == "features":
        rc = L.ovx_forward_features(C.byref(P), _ptr(xyz, C.c_double), _ptr(feat, C.c_float), r_scalar, _ptr(rad, C.c_float), N, nC, po)
    elif mode == "types":
        rc = L.ovx_forward_types(C.byref(P), _ptr(xyz, C.c_double), _ptr(types, C.c_int32), r_scalar, _ptr(rad, C.c_float), N, nC, po)
    else:
        rc = L.ovx_forward_single(C.byref(P), _ptr(xyz, C.c_double), r_scalar, _ptr(rad, C.c_float), N, po)
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return out
