"""`library='hip'` backend: the reference's Voxelizer interface on hand-written MI355X kernels.

Host layer only: argument checks (same AssertionError messages as the reference numpy backend,
molvoxel/voxelizer/numpy/voxelizer.py:171-192, 317-342, 438-455), dtype fixes (:125-130, :268-271),
channel-count inference (:275-278), RNG draws for the random transform in the reference's order
(numpy/transform.py:63-80) and the call through the C ABI (include/mvx.h). All arithmetic of the
hot path — centring, rigid transform, culls, distances, densities, accumulation — runs on the GPU.
There is no CPU fallback: construction fails without the shared library or without a HIP device.

Array arguments may be numpy arrays (host: staged through pinned memory by the library) or torch
CUDA tensors on this voxelizer's device (zero-copy via data_ptr on the current torch stream).
Grids this backend allocates are torch CUDA tensors by default (`output="torch"`), so results stay
in HBM; pass `output="numpy"` (or a numpy `out_grid`) for host arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ..contract import BaseVoxelizer
from . import _lib
from .transform import RandomTransform, draw_forward_transform

try:  # torch is plumbing (device memory + streams); the library also works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


try:  # the raw stream handle without building a torch.cuda.Stream object per call (~1.5 us -> ~0.2 us)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except Exception:  # pragma: no cover
    _raw_stream = None


def _np_isscalar(x) -> bool:
    return np.isscalar(x)


class Voxelizer(BaseVoxelizer):
    LIB = "HIP"
    transform_class = RandomTransform

    def __init__(
        self,
        resolution: float = 0.5,
        dimension: int = 64,
        radii_type: str = "scalar",
        density_type: str = "gaussian",
        precision: int = 32,
        blockdim: int | None = None,
        device=None,
        output: str = "torch",
        overlap_prepass: bool = False,
        **kwargs,
    ):
        super().__init__(resolution, dimension, radii_type, density_type, **kwargs)
        assert precision in [32, 64]
        assert output in ("torch", "numpy")
        if output == "torch" and torch is None:
            raise ImportError("output='torch' needs PyTorch; use output='numpy'")
        self.precision = precision
        self.fp = np.float32 if precision == 32 else np.float64  # numpy/voxelizer.py:34
        self._tfp = None if torch is None else (torch.float32 if precision == 32 else torch.float64)
        self.blockdim = blockdim if blockdim is not None else 8  # numpy/voxelizer.py:38
        self.num_blocks = -(-dimension // self.blockdim)
        self.output = output
        self._lib = _lib.load()
        self._device_index = self._resolve_device(device)
        self._handle = _lib.Handle()
        self._types_cache = None
        self._types_i32 = None
        self._types_fresh = False
        self._xf = _lib.MvxXform()  # reused per call: the library copies it before the call returns
        self._xf_addr = C.addressof(self._xf)
        self._has_torch_cuda = torch is not None and torch.cuda.is_available()  # asked on every call otherwise
        cfg = _lib.MvxConfig(
            float(resolution),
            float(getattr(self, "_sigma", 0.5)),
            int(dimension),
            int(self.blockdim),
            _lib.MVX_GAUSSIAN if density_type == "gaussian" else _lib.MVX_BINARY,
            self._device_index,
            int(precision),
            0,
        )
        _lib.check(self._lib.mvx_create(C.byref(cfg), C.byref(self._handle)))
        self.overlap_prepass = bool(overlap_prepass)
        if self.overlap_prepass:
            self.set_overlap_prepass(True)

    def set_overlap_prepass(self, enable: bool):
        """Loops of large `forward_batch` calls: run the pre-pass of call k+1 under the voxelize launches of call k
        (mvx_set_overlap). The caller then promises that the input tensors of a call are complete when the call is made
        (not merely ordered on the stream) and stay unchanged until its launches have run; outputs keep stream order."""
        _lib.check(self._lib.mvx_set_overlap(self._handle, 1 if enable else 0))
        self.overlap_prepass = bool(enable)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _resolve_device(device) -> int:
        if device is None or device == "cuda":
            if torch is not None and torch.cuda.is_available():
                return torch.cuda.current_device()
            return 0
        if isinstance(device, int):
            return device
        if torch is not None:
            d = torch.device(device)
            assert d.type == "cuda", "the HIP backend runs on a GPU device only"
            return d.index if d.index is not None else torch.cuda.current_device()
        raise ValueError(f"cannot interpret device={device!r}")

    @property
    def device(self):
        return torch.device("cuda", self._device_index) if torch is not None else self._device_index

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                self._lib.mvx_destroy(h)
            except Exception:
                pass
            self._handle = None

    def _density_changed(self):
        if getattr(self, "_handle", None) is not None:
            dens = _lib.MVX_GAUSSIAN if self.is_density_type_gaussian else _lib.MVX_BINARY
            _lib.check(self._lib.mvx_set_density(self._handle, dens, float(getattr(self, "_sigma", 0.5))))

    # ------------------------------------------------------------------------------------------
    # allocation / conversion (numpy/voxelizer.py:60-70, 562-583)
    def get_empty_grid(self, num_channels: int, batch_size: int | None = None, init_zero: bool = False):
        shape = self.grid_dimension(num_channels)
        if batch_size is not None:
            shape = (batch_size,) + shape
        if self.output == "torch":
            fn = torch.zeros if init_zero else torch.empty
            return fn(shape, dtype=self._tfp, device=self.device)
        if torch is not None and torch.cuda.is_available():
            # numpy grids live in pinned host memory (torch's caching host allocator): the copy back is a direct DMA
            # at PCIe speed instead of a staged pageable copy (3.4 -> ~0.7 ms per cfg-2 grid)
            fn = torch.zeros if init_zero else torch.empty
            return fn(shape, dtype=self._tfp, pin_memory=True).numpy()
        return (np.zeros if init_zero else np.empty)(shape, dtype=self.fp)

    def asarray(self, array, obj: str):
        if obj in ("coords", "center"):
            np_dt, t_dt = np.float64, "float64"
        elif obj in ("features", "radii"):
            np_dt, t_dt = (np.float32, "float32") if self.precision == 32 else (np.float64, "float64")
        elif obj == "types":
            np_dt, t_dt = np.int16, "int16"
        else:
            raise ValueError("obj should be ['coords', 'center', 'radii', types', 'features']")
        if self.output == "torch":
            if _is_torch(array):
                return array.to(device=self.device, dtype=getattr(torch, t_dt))
            return torch.as_tensor(np.asarray(array, dtype=np_dt), device=self.device)
        if _is_torch(array):
            array = array.detach().cpu().numpy()
        return np.asarray(array, dtype=np_dt)

    def to(self, device):
        """torch-backend compatibility: a handle is bound to one GPU; moving re-creates it."""
        idx = self._resolve_device(device)
        if idx != self._device_index:
            kw = {"sigma": self._sigma} if self.is_density_type_gaussian else {}
            return type(self)(self._resolution, self._dimension, self._radii_type, self._density_type, self.precision,
                              self.blockdim, idx, self.output, self.overlap_prepass, **kw)
        return self

    def cuda(self):
        return self

    def cpu(self):
        raise NotImplementedError("the HIP backend has no CPU path; use the upstream numpy backend on the host")

    # ------------------------------------------------------------------------------------------
    # argument plumbing
    def _stream(self):
        """The caller's current torch stream as a hipStream_t value (0 = the null stream)."""
        if self._has_torch_cuda:
            if _raw_stream is not None:
                return _raw_stream(self._device_index)
            return torch.cuda.current_stream(self._device_index).cuda_stream
        return 0

    def _on_device(self, x) -> bool:
        return _is_torch(x) and x.is_cuda and x.device.index == self._device_index

    def _prepare_inputs(self, coords, chan, chan_kind, radii):
        """Returns (coords, chan, radii_array_or_None, in_kind, keepalive). chan_kind in {features, types, None}."""
        dev = self._on_device(coords)
        keep = []
        if dev:
            c = coords if (coords.dtype == torch.float64 and coords.is_contiguous()) else coords.to(torch.float64).contiguous()
            ch = None
            if chan_kind == "features":
                ch = (chan if _is_torch(chan) else torch.as_tensor(np.asarray(chan), device=self.device)).to(
                    device=self.device, dtype=self._tfp).contiguous()
            elif chan_kind == "types":
                t = chan if _is_torch(chan) else torch.as_tensor(np.asarray(chan), device=self.device)
                ch = self._types_as_int32(t)
            r = None
            if not _np_isscalar(radii):
                r = (radii if _is_torch(radii) else torch.as_tensor(np.asarray(radii), device=self.device)).to(
                    device=self.device, dtype=self._tfp).contiguous()
            keep += [c, ch, r]
            return c, ch, r, _lib.MVX_DEVICE, keep
        if _is_torch(coords):
            coords = coords.detach().cpu().numpy()
        c = np.ascontiguousarray(coords, dtype=np.float64)
        ch = None
        if chan_kind == "features":
            ch = chan.detach().cpu().numpy() if _is_torch(chan) else np.asarray(chan)
            ch = np.ascontiguousarray(ch, dtype=self.fp)
        elif chan_kind == "types":
            t = chan.detach().cpu().numpy() if _is_torch(chan) else np.asarray(chan)
            ch = np.ascontiguousarray(t.astype(np.int16), dtype=np.int32)
        r = None
        if not _np_isscalar(radii):
            r = radii.detach().cpu().numpy() if _is_torch(radii) else np.asarray(radii)
            r = np.ascontiguousarray(r, dtype=self.fp)
        keep += [c, ch, r]
        return c, ch, r, _lib.MVX_HOST, keep

    @staticmethod
    def _ptr(x):
        """Address for a `void *` argument (ctypes takes None / int)."""
        if x is None:
            return None
        if _is_torch(x):
            return x.data_ptr()
        return x.ctypes.data

    def _make_xform(self, center, random_translation, random_rotation, on_device=False, keep=None):
        """Address of one mvx_xform (the library copies it during the call): centring + the random transform drawn in
        the reference's RNG order. A `center` tensor on this device is handed over by pointer (MVX_XF_CENTER_PTR) when
        the coordinates live there too: its value never visits the host, so the call does not synchronise."""
        xf = self._xf
        flags = 0
        if center is not None:
            if on_device and self._on_device(center):
                cen = center if (center.dtype == torch.float64 and center.is_contiguous()) else center.to(torch.float64).contiguous()
                assert cen.numel() == 3, "center should be Array[3,]"
                if keep is not None:
                    keep.append(cen)
                xf.center_ptr = cen.data_ptr()
                flags |= _lib.MVX_XF_CENTER | _lib.MVX_XF_CENTER_PTR
            else:
                cen = center.detach().cpu().numpy() if _is_torch(center) else np.asarray(center)
                cen = cen.reshape(3).astype(np.float64)
                xf.center[0], xf.center[1], xf.center[2] = float(cen[0]), float(cen[1]), float(cen[2])
                flags |= _lib.MVX_XF_CENTER
        if random_rotation or (random_translation is not None and random_translation > 0.0):
            translation, quaternion = draw_forward_transform(random_translation, random_rotation)
            if quaternion is not None:
                xf.quat[0], xf.quat[1], xf.quat[2], xf.quat[3] = (float(q) for q in quaternion)
                flags |= _lib.MVX_XF_ROTATE
            if translation is not None:
                t = translation.reshape(3)
                xf.trans[0], xf.trans[1], xf.trans[2] = float(t[0]), float(t[1]), float(t[2])
                flags |= _lib.MVX_XF_TRANSLATE
        xf.flags = flags
        return self._xf_addr if flags else None

    def _resolve_out(self, out_grid, shape):
        """Returns (buffer passed to the library, out_kind, object to return)."""
        if out_grid is None:
            out_grid = self.get_empty_grid(shape[0])
        if _is_torch(out_grid):
            if self._on_device(out_grid) and out_grid.is_contiguous() and out_grid.dtype == self._tfp:
                return out_grid, _lib.MVX_DEVICE, out_grid, None
            tmp = torch.empty(tuple(out_grid.shape), dtype=self._tfp, device=self.device)
            return tmp, _lib.MVX_DEVICE, out_grid, "copy_torch"
        if out_grid.flags.c_contiguous and out_grid.dtype == self.fp:
            return out_grid, _lib.MVX_HOST, out_grid, None
        tmp = np.empty(out_grid.shape, dtype=self.fp)
        return tmp, _lib.MVX_HOST, out_grid, "copy_numpy"

    @staticmethod
    def _finish_out(buf, ret, how):
        if how == "copy_torch":
            ret.copy_(buf)
        elif how == "copy_numpy":
            ret[...] = buf
        return ret

    def _radii_type_code(self):
        if self.is_radii_type_scalar:
            return _lib.MVX_RADII_SCALAR
        if self.is_radii_type_atom_wise:
            return _lib.MVX_RADII_ATOM
        return _lib.MVX_RADII_CHANNEL

    # ------------------------------------------------------------------------------------------
    # VECTOR  (replaces numpy/voxelizer.py:97-236)
    def forward_features(self, coords, center, features, radii, random_translation=0.0, random_rotation=False,
                         out_grid=None):
        """coords (V,3), center (3,) | None, features (V,C), radii scalar | (V,) | (C,); out (C,D,H,W)."""
        self._check_args_features(coords, features, radii, out_grid)
        C_ = features.shape[1]
        c, f, r, in_kind, keep = self._prepare_inputs(coords, features, "features", radii)
        xf = self._make_xform(center, random_translation, random_rotation, in_kind == _lib.MVX_DEVICE, keep)
        buf, out_kind, ret, how = self._resolve_out(out_grid, (C_,))
        rs = float(radii) if r is None else 0.0
        rc = self._lib.mvx_forward_features(
            self._handle, self._ptr(c), self._ptr(f), self._ptr(r), rs, self._radii_type_code(), c.shape[0], C_,
            xf, self._ptr(buf), in_kind, out_kind, self._stream())
        if rc:
            _lib.check(rc)
        return self._finish_out(buf, ret, how) if how else ret

    def _check_args_features(self, coords, features, radii, out_grid=None):
        V = coords.shape[0]
        C_ = features.shape[1]
        D = H = W = self.dimension
        assert features.shape[0] == V, f"atom features does not match number of atoms: {features.shape[0]} vs {V}"
        assert features.ndim == 2, f"atom features does not match dimension: {features.shape} vs {(V,'*')}"
        if self.is_radii_type_scalar:
            assert _np_isscalar(radii), "the radii type of voxelizer is `scalar`, radii should be scalar"
        elif self.is_radii_type_channel_wise:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `channel-wise`, radii should be Array[{C_},]"
            assert tuple(radii.shape) == (C_,), f"radii does not match dimension (number of channels,): {tuple(radii.shape)} vs {(C_,)}"
        else:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `atom-wise`, radii should be Array[{V},]"
            assert tuple(radii.shape) == (V,), f"radii does not match dimension (number of atoms,): {tuple(radii.shape)} vs {(V,)}"
        if out_grid is not None:
            assert tuple(out_grid.shape) == (C_, D, H, W), f"Output grid dimension incorrect: {tuple(out_grid.shape)} vs {(C_,D,H,W)}"

    # ------------------------------------------------------------------------------------------
    # INDEX  (replaces numpy/voxelizer.py:240-366)
    def forward_types(self, coords, center, types, radii, random_translation=0.0, random_rotation=False,
                      out_grid=None):
        """coords (V,3), center (3,) | None, types (V,), radii scalar | (V,) | (C,); out (C,D,H,W)."""
        n_types = self._check_args_types(coords, types, radii, out_grid)
        if out_grid is not None:
            C_ = out_grid.shape[0]  # extra channels stay zero (numpy/voxelizer.py:337)
        elif self.is_radii_type_channel_wise:
            C_ = radii.shape[0]  # numpy/voxelizer.py:275-276
        else:
            C_ = n_types  # max(types) + 1 over ALL atoms, numpy/voxelizer.py:278
        c, t, r, in_kind, keep = self._prepare_inputs(coords, types, "types", radii)
        if self.is_radii_type_channel_wise and r is not None and r.shape[0] < C_:
            # channel-wise radii are indexed by type only; pad so the (C,) contract of the ABI holds
            pad = C_ - r.shape[0]
            r = torch.cat([r, r.new_ones(pad)]) if _is_torch(r) else np.concatenate([r, np.ones(pad, self.fp)])
        xf = self._make_xform(center, random_translation, random_rotation, in_kind == _lib.MVX_DEVICE, keep)
        buf, out_kind, ret, how = self._resolve_out(out_grid, (C_,))
        rs = float(radii) if r is None else 0.0
        rc = self._lib.mvx_forward_types(
            self._handle, self._ptr(c), self._ptr(t), self._ptr(r), rs, self._radii_type_code(), c.shape[0], int(C_),
            xf, self._ptr(buf), in_kind, out_kind, self._stream())
        if rc:
            _lib.check(rc)
        return self._finish_out(buf, ret, how) if how else ret

    def _types_as_int32(self, t):
        """Device types in the ABI's int32, through int16 like the reference's cast (numpy/voxelizer.py:269).
        The converted tensor is remembered while the same unmodified tensor keeps coming (two cast kernels per call
        otherwise)."""
        hit = self._types_i32
        self._types_fresh = False
        if hit is None or hit[0] is not t or hit[1] != t._version:
            # (the source tensor is held, not its address: a freed tensor's memory can come back with other content)
            hit = self._types_i32 = (t, t._version, t.to(device=self.device).to(torch.int16).to(torch.int32).contiguous())
            self._types_fresh = True  # converted on the current stream a moment ago
        return hit[2]

    def _types_extent(self, types):
        """(min, max) of the type indices. For a device tensor this costs a kernel and a synchronisation, so the
        answer is remembered for as long as the same tensor is passed unmodified (torch bumps `_version` on every
        in-place write) - the per-molecule loop of test/test_time_numpy.py:11-15 asks thousands of times.
        Limitation: a write that bypasses torch's version counter (`.data`, raw pointers, DLPack consumers, this
        library's own mvx_memcpy) is not seen; such callers should pass a fresh tensor (or clone) after writing."""
        if not _is_torch(types):
            return int(types.min()), int(types.max())
        hit = self._types_cache
        if hit is None or hit[0] is not types or hit[1] != types._version:
            lo, hi = torch.aminmax(types)
            hit = self._types_cache = (types, types._version, int(lo), int(hi))
        return hit[2], hit[3]

    def _check_args_types(self, coords, types, radii, out_grid=None):
        V = coords.shape[0]
        tmin, tmax = self._types_extent(types)
        C_ = tmax + 1
        D = H = W = self.dimension
        assert tuple(types.shape) == (V,), f"types does not match dimension: {tuple(types.shape)} vs {(V,)}"
        assert tmin >= 0, "types must be non-negative channel indices"
        if self.is_radii_type_scalar:
            assert _np_isscalar(radii), "the radii type of voxelizer is `scalar`, radii should be scalar"
        elif self.is_radii_type_channel_wise:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `channel-wise`, radii should be Array[{C_},]"
            assert tuple(radii.shape) == (C_,), f"radii does not match dimension (number of channels,): {tuple(radii.shape)} vs {(C_,)}"
        else:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `atom-wise`, radii should be Array[{V},]"
            assert tuple(radii.shape) == (V,), f"radii does not match dimension (number of atoms,): {tuple(radii.shape)} vs {(V,)}"
        if out_grid is not None:
            assert out_grid.shape[0] >= C_, f"Output channel is less than number of types: {out_grid.shape[0]} < {C_}"
            assert tuple(out_grid.shape[1:]) == (D, H, W), f'Output grid dimension incorrect: {tuple(out_grid.shape)} vs {("*",D,H,W)}'
        return C_

    # ------------------------------------------------------------------------------------------
    # SINGLE  (replaces numpy/voxelizer.py:370-477)
    def forward_single(self, coords, center, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        """coords (V,3), center (3,) | None, radii scalar | (V,); out (1,D,H,W)."""
        self._check_args_single(coords, radii, out_grid)
        c, _, r, in_kind, keep = self._prepare_inputs(coords, None, None, radii)
        xf = self._make_xform(center, random_translation, random_rotation, in_kind == _lib.MVX_DEVICE, keep)
        buf, out_kind, ret, how = self._resolve_out(out_grid, (1,))
        rs = float(radii) if r is None else 0.0
        rc = self._lib.mvx_forward_single(
            self._handle, self._ptr(c), self._ptr(r), rs, self._radii_type_code(), c.shape[0],
            xf, self._ptr(buf), in_kind, out_kind, self._stream())
        if rc:
            _lib.check(rc)
        return self._finish_out(buf, ret, how) if how else ret

    def _check_args_single(self, coords, radii, out_grid=None):
        V = coords.shape[0]
        D = H = W = self.dimension
        assert not self.is_radii_type_channel_wise, "Channel-Wise Radii Type is not supported"
        if self.is_radii_type_scalar:
            assert _np_isscalar(radii), "the radii type of voxelizer is `scalar`, radii should be scalar"
        else:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `atom-wise`, radii should be Array[{V},]"
            assert tuple(radii.shape) == (V,), f"radii does not match dimension (number of atoms,): {tuple(radii.shape)} vs {(V,)}"
        if out_grid is not None:
            assert out_grid.shape[0] == 1, "Output channel should be 1"
            assert tuple(out_grid.shape[1:]) == (D, H, W), f'Output grid dimension incorrect: {tuple(out_grid.shape)} vs {("*",D,H,W)}'

    # ------------------------------------------------------------------------------------------
    # BATCH (the loop of test/test_time_numpy.py:11-15 as one launch; molecules are independent)
    def forward_batch(self, coords, offsets, centers, channels, radii, num_channels=None, out_grid=None,
                      random_translation=0.0, random_rotation=False):
        """Voxelize B molecules stored back to back.

        coords (sumN,3) float64; offsets (B+1,) int64 (host); centers (B,3) | None;
        channels: (sumN,C) float -> features, (sumN,) int -> types, None -> single;
        radii: python float | (sumN,) | (C,) per this voxelizer's radii_type.
        out_grid: (B,C,D,H,W) float32 (torch CUDA tensor on this device or numpy), fully overwritten.
        A random transform, if requested, is drawn per molecule in molecule order.
        """
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        B = offsets.shape[0] - 1
        assert offsets[0] == 0 and offsets[-1] == coords.shape[0], "offsets must span coords"
        if channels is None:
            kind, C_ = None, 1
        elif channels.ndim == 1:
            kind = "types"
            C_ = num_channels if num_channels is not None else (
                radii.shape[0] if self.is_radii_type_channel_wise else int(channels.max()) + 1)
        else:
            kind, C_ = "features", channels.shape[1]
        self._check_args_batch(coords, channels, kind, radii, int(C_))
        c, ch, r, in_kind, keep = self._prepare_inputs(coords, channels, kind, radii)
        # With overlap_prepass the library's side stream reads the inputs without waiting for the caller's stream.
        # Arrays this layer had to convert just now (dtype / layout fixes, the int32 copy of `types`) were produced
        # ON that stream a moment ago: make them complete first (first call with a given tensor only: the copies are
        # cached or the caller passes the right dtype)
        fresh = self.overlap_prepass and in_kind == _lib.MVX_DEVICE and (
            c is not coords or (kind == "features" and ch is not channels) or (r is not None and r is not radii)
            or (kind == "types" and self._types_fresh))
        if kind == "types" and self.is_radii_type_channel_wise and r.shape[0] < C_:
            # channel-wise radii are indexed by type only; pad so the (C,) contract of the ABI holds (as forward_types)
            pad = int(C_) - r.shape[0]
            r = torch.cat([r, r.new_ones(pad)]) if _is_torch(r) else np.concatenate([r, np.ones(pad, self.fp)])
            fresh = fresh or self.overlap_prepass
        need_xf = centers is not None or random_rotation or (random_translation and random_translation > 0.0)
        xf_ptr = None
        if need_xf:
            xfs = (_lib.MvxXform * B)()
            cen = dev_cen = None
            if centers is not None:
                if in_kind == _lib.MVX_DEVICE and self._on_device(centers):  # by pointer: no copy to the host
                    dev_cen = centers.to(torch.float64).contiguous().reshape(B, 3)
                    keep.append(dev_cen)
                    fresh = fresh or (self.overlap_prepass and dev_cen.data_ptr() != centers.data_ptr())
                else:
                    cen = centers.detach().cpu().numpy() if _is_torch(centers) else np.asarray(centers)
                    cen = cen.reshape(B, 3)
            for b in range(B):
                self._make_xform(None if cen is None else cen[b], random_translation, random_rotation)
                if dev_cen is not None:
                    self._xf.center_ptr = dev_cen.data_ptr() + 24 * b
                    self._xf.flags |= _lib.MVX_XF_CENTER | _lib.MVX_XF_CENTER_PTR
                C.memmove(C.addressof(xfs) + b * C.sizeof(_lib.MvxXform), self._xf_addr, C.sizeof(_lib.MvxXform))
            xf_ptr = C.addressof(xfs)
        if out_grid is None:
            out_grid = self.get_empty_grid(C_, batch_size=B)
        assert tuple(out_grid.shape) == (B,) + self.grid_dimension(C_), (
            f"Output grid dimension incorrect: {tuple(out_grid.shape)} vs {(B,) + self.grid_dimension(C_)}")
        buf, out_kind, ret, how = self._resolve_out(out_grid, None)
        if fresh:
            torch.cuda.current_stream(self._device_index).synchronize()
        rs = float(radii) if _np_isscalar(radii) else 0.0
        off_ptr = offsets.ctypes.data
        rt = self._radii_type_code()
        if kind == "features":
            rc = self._lib.mvx_forward_features_batch(self._handle, self._ptr(c), self._ptr(ch), self._ptr(r), rs, rt,
                                                      off_ptr, xf_ptr, B, int(C_), self._ptr(buf), in_kind, out_kind,
                                                      self._stream())
        elif kind == "types":
            rc = self._lib.mvx_forward_types_batch(self._handle, self._ptr(c), self._ptr(ch), self._ptr(r), rs, rt,
                                                   off_ptr, xf_ptr, B, int(C_), self._ptr(buf), in_kind, out_kind,
                                                   self._stream())
        else:
            assert not self.is_radii_type_channel_wise, "Channel-Wise Radii Type is not supported"
            rc = self._lib.mvx_forward_single_batch(self._handle, self._ptr(c), self._ptr(r), rs, rt, off_ptr, xf_ptr,
                                                    B, self._ptr(buf), in_kind, out_kind, self._stream())
        _lib.check(rc)
        return self._finish_out(buf, ret, how)

    def _check_args_batch(self, coords, channels, kind, radii, C_):
        """The per-molecule checks (_check_args_features / _types / _single) for atoms stored back to back: the
        library reads sumN rows of channels and sumN | C radii, so every array must really have them."""
        V = coords.shape[0]
        assert coords.ndim == 2 and coords.shape[1] == 3, f"coords does not match dimension: {tuple(coords.shape)} vs {(V, 3)}"
        if kind == "features":
            assert channels.shape[0] == V, f"atom features does not match number of atoms: {channels.shape[0]} vs {V}"
        elif kind == "types":
            assert tuple(channels.shape) == (V,), f"types does not match dimension: {tuple(channels.shape)} vs {(V,)}"
        else:
            assert not self.is_radii_type_channel_wise, "Channel-Wise Radii Type is not supported"
        if self.is_radii_type_scalar:
            assert _np_isscalar(radii), "the radii type of voxelizer is `scalar`, radii should be scalar"
        elif self.is_radii_type_channel_wise:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `channel-wise`, radii should be Array[{C_},]"
            if kind == "features":
                assert tuple(radii.shape) == (C_,), f"radii does not match dimension (number of channels,): {tuple(radii.shape)} vs {(C_,)}"
            else:  # types: radii are gathered by type; every type below num_channels needs one
                assert radii.ndim == 1 and 0 < radii.shape[0] <= C_, f"radii does not match dimension (number of channels,): {tuple(radii.shape)} vs {(C_,)}"
                if V > 0:
                    tmax = int(channels.max())
                    assert tmax < radii.shape[0], f"radii does not match dimension (number of channels,): {tuple(radii.shape)} vs {(tmax + 1,)}"
        else:
            assert not _np_isscalar(radii), f"the radii type of voxelizer is `atom-wise`, radii should be Array[{V},]"
            assert tuple(radii.shape) == (V,), f"radii does not match dimension (number of atoms,): {tuple(radii.shape)} vs {(V,)}"

    # ------------------------------------------------------------------------------------------
    # measurement hooks used by bench.py (HIP events around the voxelize kernel on the launch stream)
    def debug_option(self, name: str, value: int):
        """Testing aid (mvx_debug_set_option): force code paths production sizes rarely reach."""
        _lib.check(self._lib.mvx_debug_set_option(self._handle, name.encode(), int(value)))

    def set_profiling(self, enable: bool):
        _lib.check(self._lib.mvx_set_profiling(self._handle, 1 if enable else 0))

    def read_kernel_times_ms(self):
        """Durations (ms) of the voxelize kernel for every launch since profiling was enabled / last read."""
        buf = (C.c_float * 1024)()
        n = C.c_int32(0)
        _lib.check(self._lib.mvx_profile_read(self._handle, buf, 1024, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0.0)
        _lib.check(self._lib.mvx_last_kernel_ms(self._handle, C.byref(ms)))
        return ms.value

    @staticmethod
    def do_random_transform(coords, center, random_translation, random_rotation):
        from .transform import do_random_transform

        return do_random_transform(coords, center, random_translation, random_rotation)


def transform_on_device(coords, center, translation, quaternion):
    """do_transform for a torch CUDA tensor (N,3): runs mvx_transform_coords on the tensor's device."""
    lib = _lib.load()
    dev = coords.device.index if coords.device.index is not None else torch.cuda.current_device()
    vox = _transform_handles.get(dev)
    if vox is None:
        vox = _transform_handles[dev] = Voxelizer(0.5, 8, device=dev)
    xf = _lib.MvxXform()
    flags = 0
    if quaternion is not None:
        xf.quat[:] = [float(q) for q in quaternion]
        flags |= _lib.MVX_XF_ROTATE
        if center is not None:
            cen = center.detach().cpu().numpy() if _is_torch(center) else np.asarray(center)
            xf.center[:] = cen.reshape(3).astype(np.float64).tolist()
            flags |= _lib.MVX_XF_CENTER | _lib.MVX_XF_RECENTER
    if translation is not None:
        tr = translation.detach().cpu().numpy() if _is_torch(translation) else np.asarray(translation)
        xf.trans[:] = tr.reshape(3).astype(np.float32).tolist()
        flags |= _lib.MVX_XF_TRANSLATE
    xf.flags = flags
    src = coords.to(torch.float64).contiguous()
    out = torch.empty_like(src)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.mvx_transform_coords(vox._handle, src.data_ptr(), src.shape[0], C.addressof(xf), out.data_ptr(),
                                        _lib.MVX_DEVICE, _lib.MVX_DEVICE, stream))
    return out


_transform_handles: dict = {}
