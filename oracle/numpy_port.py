"""numpy/scipy port of the reference numpy backend's block algorithm.

TEST INFRASTRUCTURE ONLY (checker + `cpu_baseline` in bench.py). Never imported by
molvoxel_amd/. The reference package cannot travel to the GPU box, so this port is what
gets timed there as the "numpy CPU path"; its equality with the imported reference
(results: bit-identical for binary and Gaussian types/single, <=2.4e-7 for Gaussian
features; speed: same calls, same shapes) is established by tests/test_oracle_golden.py
and oracle/gen_golden.py in the build container.

Follows (file:line relative to /root/reference):
  block set-up, axis, bounds ........ molvoxel/voxelizer/numpy/voxelizer.py:37-58
  forward_features/types/single ..... molvoxel/voxelizer/numpy/voxelizer.py:97-169, 240-315, 370-436
  per-block kernels ................. molvoxel/voxelizer/numpy/voxelizer.py:194-236, 344-366, 457-477
  culls ............................. molvoxel/voxelizer/numpy/voxelizer.py:481-527
  cdist -> float32 -> divide -> density: molvoxel/voxelizer/numpy/voxelizer.py:531-560

Same numerical recipe per reference block (scipy cdist in fp64, cast to float32, divide by
the radius, exp / <=, BLAS matmul), organised differently: one function, boolean
per-axis admission tables instead of dicts of index arrays, grid points by broadcasting.
Coordinates are expected AFTER centring and random transform.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.spatial.distance import cdist


class GridSpec:
    """Geometry cache: axis, reference block slices, bounds."""

    def __init__(self, resolution=0.5, dimension=64, blockdim=None):
        self.resolution = resolution
        self.dimension = dimension
        self.blockdim = 8 if blockdim is None else blockdim
        self.width = resolution * (dimension - 1)
        self.upper = self.width / 2.0
        self.lower = -1 * self.upper
        self.axis = np.arange(dimension, dtype=np.float64) * resolution - (self.width / 2.0)
        self.nb = math.ceil(dimension / self.blockdim)
        self.slices = [slice(b * self.blockdim, min((b + 1) * self.blockdim, dimension)) for b in range(self.nb)]
        self.bounds = [self.axis[b * self.blockdim] + (resolution / 2.0) for b in range(1, self.nb)]
        self._pts = {}

    def points(self, bx, by, bz):
        key = (bx, by, bz)
        pts = self._pts.get(key)
        if pts is None:
            ax, ay, az = self.axis[self.slices[bx]], self.axis[self.slices[by]], self.axis[self.slices[bz]]
            pts = np.empty((ax.size, ay.size, az.size, 3), dtype=np.float64)
            pts[..., 0] = ax[:, None, None]
            pts[..., 1] = ay[None, :, None]
            pts[..., 2] = az[None, None, :]
            pts = self._pts[key] = pts.reshape(-1, 3)
        return pts


def _box_keep(spec, xyz, size):
    if np.isscalar(size):
        lo_ok = (xyz > spec.lower - size).all(axis=1)
        hi_ok = (xyz < spec.upper + size).all(axis=1)
    else:
        s = size[:, None]
        lo_ok = (xyz + s > spec.lower).all(axis=1)
        hi_ok = (xyz - s < spec.upper).all(axis=1)
    return np.flatnonzero(lo_ok & hi_ok)


def _axis_tables(spec, xyz, size):
    """(3, nb, V) bool: atom admitted to block b along each axis (strict compares, fp64)."""
    V = xyz.shape[0]
    tab = np.ones((3, spec.nb, V), dtype=bool)
    for a in range(3):
        p = xyz[:, a]
        for b in range(spec.nb):
            if b >= 1:
                tab[a, b] &= p > spec.bounds[b - 1] - size
            if b <= spec.nb - 2:
                tab[a, b] &= p < spec.bounds[b] + size
    return tab


def _density(dist32, radius, density, sigma):
    """dist32: (V, P) in the value type; radius: python float or (V, 1) in the value type."""
    dr = np.divide(dist32, radius)
    if density == "binary":
        return np.less_equal(dr, 1.0, dr)
    val = np.exp(-0.5 * ((dr / sigma) ** 2))
    val[dr > 1.0] = 0
    return val


def voxelize(
    spec: GridSpec,
    coords,
    channels,
    radii,
    *,
    radii_type="scalar",
    density="gaussian",
    sigma=0.5,
    out=None,
    num_channels=None,
    precision=32,
):
    """channels: None (single), int (V,) (types) or float (V, C) (features).
    precision: 32 | 64, the reference's `self.fp` (numpy/voxelizer.py:33-34)."""
    fp = np.float32 if precision == 32 else np.float64  # the value type everything below is cast to
    D = spec.dimension
    xyz = np.asarray(coords, dtype=np.float64)
    mode = "single" if channels is None else ("types" if np.ndim(channels) == 1 else "features")
    if not np.isscalar(radii):
        radii = np.asarray(radii).astype(fp, copy=False)
    if mode == "features":
        chan = np.asarray(channels).astype(fp, copy=False)
        C = chan.shape[1]
    elif mode == "types":
        chan = np.asarray(channels).astype(np.int16, copy=False)
        if num_channels is not None:
            C = num_channels
        elif radii_type == "channel-wise":
            C = radii.shape[0]
        else:
            C = int(chan.max()) + 1
    else:
        chan, C = None, 1
    if out is None:
        out = np.empty((C, D, D, D), dtype=fp)
    if mode != "features":
        out.fill(0.0)

    chanwise_feat = mode == "features" and radii_type == "channel-wise"
    if mode == "types" and radii_type == "channel-wise":
        radii = radii[chan]
    size = radii.max() if chanwise_feat else radii  # np.float32 scalar for channel-wise features

    keep = _box_keep(spec, xyz, size)
    xyz = xyz[keep]
    if chan is not None:
        chan = chan[keep]
    per_atom = (not np.isscalar(radii)) and not chanwise_feat
    if per_atom:
        radii = radii[keep]
        size = radii

    if spec.nb > 1:
        tab = _axis_tables(spec, xyz, size if np.isscalar(size) else size)
    for bx in range(spec.nb):
        for by in range(spec.nb):
            for bz in range(spec.nb):
                view = out[:, spec.slices[bx], spec.slices[by], spec.slices[bz]]
                if spec.nb > 1:
                    idx = np.flatnonzero(tab[0, bx] & tab[1, by] & tab[2, bz])
                else:
                    idx = np.arange(xyz.shape[0])
                if idx.size == 0:
                    if mode == "features":
                        view.fill(0.0)
                    continue
                shp = view.shape[1:]
                dist = cdist(xyz[idx], spec.points(bx, by, bz)).astype(fp)
                if chanwise_feat:
                    for c in range(C):
                        val = _density(dist, radii[c], density, sigma)
                        view[c] = np.matmul(chan[idx, c], val).reshape(shp)
                    continue
                rad = radii[idx][:, None] if per_atom else radii
                val = _density(dist, rad, density, sigma)
                if mode == "features":
                    view[:] = np.matmul(chan[idx].T, val).reshape((-1,) + shp)
                elif mode == "types":
                    val = val.reshape((-1,) + shp)
                    for row, t in zip(val, chan[idx]):
                        view[t] += row
                else:
                    view[0] = val.sum(axis=0).reshape(shp)
    return out
