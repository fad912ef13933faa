"""ctypes binding of libmvx_hip.so (the C ABI declared in include/mvx.h).

There is deliberately no fallback: if the shared library is missing or cannot be loaded the
import of the HIP backend fails loudly. Build it with `make -C molvoxel_amd/csrc` (or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "csrc")
LIB_PATH = os.path.join(CSRC, "libmvx_hip.so")

MVX_HOST, MVX_DEVICE = 0, 1
MVX_GAUSSIAN, MVX_BINARY = 0, 1
MVX_RADII_SCALAR, MVX_RADII_ATOM, MVX_RADII_CHANNEL = 0, 1, 2
MVX_XF_CENTER, MVX_XF_ROTATE, MVX_XF_TRANSLATE, MVX_XF_RECENTER, MVX_XF_CENTER_PTR = 1, 2, 4, 8, 16


class MvxConfig(C.Structure):
    _fields_ = [
        ("resolution", C.c_double),
        ("sigma", C.c_double),
        ("dimension", C.c_int32),
        ("blockdim", C.c_int32),
        ("density", C.c_int32),
        ("device", C.c_int32),
        ("precision", C.c_int32),
        ("reserved", C.c_int32),
    ]


class MvxXform(C.Structure):
    _fields_ = [
        ("center", C.c_double * 3),
        ("quat", C.c_double * 4),
        ("trans", C.c_float * 3),
        ("flags", C.c_uint32),
        ("center_ptr", C.c_void_p),
    ]


class MvxPlanQuery(C.Structure):
    _fields_ = [
        ("dimension", C.c_int32),
        ("blockdim", C.c_int32),
        ("precision", C.c_int32),
        ("mode", C.c_int32),
        ("radii_type", C.c_int32),
        ("B", C.c_int32),
        ("C", C.c_int32),
        ("out_aligned16", C.c_int32),
        ("total_atoms", C.c_int64),
        ("max_atoms", C.c_int64),
    ]


class MvxPlan(C.Structure):
    _fields_ = [(name, C.c_int32) for name in (
        "route", "nsx", "nsy", "nzc", "nw", "ct", "ncc", "nfull", "ct_rem", "nchunk", "pace", "grouped", "lane_range",
        "vec_store", "xcd_ranges", "cpad", "weights_in_place", "reserved")]


MVX_ROUTE_BINNED, MVX_ROUTE_DIRECT, MVX_ROUTE_F64_DENSE, MVX_ROUTE_F64_MX = 0, 1, 2, 3
MODES = {"features": 0, "types": 1, "single": 2}
RADII = {"scalar": MVX_RADII_SCALAR, "atom-wise": MVX_RADII_ATOM, "channel-wise": MVX_RADII_CHANNEL}

Handle = C.c_void_p
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); pointers to caller data are passed as integers (void*)
SIGNATURES = {
    "mvx_version": (C.c_int, []),
    "mvx_last_error": (C.c_char_p, []),
    "mvx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mvx_create": (C.c_int, [C.POINTER(MvxConfig), C.POINTER(Handle)]),
    "mvx_destroy": (C.c_int, [Handle]),
    "mvx_set_density": (C.c_int, [Handle, _i32, _dbl]),
    "mvx_set_overlap": (C.c_int, [Handle, _i32]),
    "mvx_forward_features_batch": (C.c_int, [Handle, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "mvx_forward_types_batch": (C.c_int, [Handle, _vp, _vp, _vp, _dbl, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "mvx_forward_single_batch": (C.c_int, [Handle, _vp, _vp, _dbl, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "mvx_forward_features": (C.c_int, [Handle, _vp, _vp, _vp, _dbl, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _vp]),
    "mvx_forward_types": (C.c_int, [Handle, _vp, _vp, _vp, _dbl, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _vp]),
    "mvx_forward_single": (C.c_int, [Handle, _vp, _vp, _dbl, _i32, _i64, _vp, _vp, _i32, _i32, _vp]),
    "mvx_transform_coords": (C.c_int, [Handle, _vp, _i64, _vp, _vp, _i32, _i32, _vp]),
    "mvx_set_profiling": (C.c_int, [Handle, _i32]),
    "mvx_profile_read": (C.c_int, [Handle, C.POINTER(C.c_float), _i32, C.POINTER(_i32)]),
    "mvx_last_kernel_ms": (C.c_int, [Handle, C.POINTER(C.c_float)]),
    "mvx_debug_read_records": (C.c_int, [Handle, _vp, _i64, _vp]),
    "mvx_debug_set_option": (C.c_int, [Handle, C.c_char_p, _i32]),
    "mvx_plan_call": (C.c_int, [C.POINTER(MvxPlanQuery), C.POINTER(MvxPlan)]),
    "mvx_alloc": (C.c_int, [Handle, _i64, C.POINTER(C.c_void_p)]),
    "mvx_free": (C.c_int, [Handle, _vp]),
    "mvx_memcpy": (C.c_int, [Handle, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mvx_memset_zero": (C.c_int, [Handle, _vp, _i64, _vp]),
    "mvx_stream_sync": (C.c_int, [Handle, _vp]),
}

_lib = None


def load():
    """Load libmvx_hip.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP voxelizer has no CPU fallback. "
            "Build it with `make -C molvoxel_amd/csrc` (needs hipcc, --offload-arch=gfx950)."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().mvx_last_error()
        raise RuntimeError(f"libmvx_hip error {rc}: {msg.decode() if msg else ''}")


def plan_call(dimension, C_, B=1, total_atoms=0, max_atoms=None, mode="features", radii_type="scalar", precision=32,
              blockdim=8, out_aligned16=True) -> dict:
    """How libmvx_hip would execute a call of this shape (mvx_plan_call: a pure host function, no GPU needed)."""
    q = MvxPlanQuery(dimension, blockdim, precision, MODES[mode], RADII[radii_type], B, C_, 1 if out_aligned16 else 0,
                     total_atoms, total_atoms if max_atoms is None else max_atoms)
    p = MvxPlan()
    check(load().mvx_plan_call(C.byref(q), C.byref(p)))
    return {name: getattr(p, name) for name, _ in MvxPlan._fields_ if name != "reserved"}


def device_count() -> int:
    n = C.c_int(0)
    rc = load().mvx_device_count(C.byref(n))
    return n.value if rc == 0 else 0
