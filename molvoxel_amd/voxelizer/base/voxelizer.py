"""Abstract operator contract of a voxelizer backend.

Mirror of the reference's BaseVoxelizer (molvoxel/voxelizer/base/voxelizer.py:9-176): same
attribute / property / method names, argument meaning and dispatch rule, so that code written
against `molvoxel.Voxelizer` runs unchanged against the HIP backend.
"""
from __future__ import annotations

import abc

import numpy as np

from .transform import BaseRandomTransform


class BaseVoxelizer(abc.ABC):
    LIB = None
    transform_class: type = BaseRandomTransform
    RADII_TYPE_LIST = ["scalar", "channel-wise", "atom-wise"]  # base/voxelizer.py:12
    DENSITY_TYPE_LIST = ["gaussian", "binary"]  # base/voxelizer.py:13

    def __init__(self, resolution=0.5, dimension=48, radii_type="scalar", density_type="gaussian", **kwargs):
        assert radii_type in self.RADII_TYPE_LIST
        assert density_type in self.DENSITY_TYPE_LIST
        self._resolution = resolution
        self._dimension = dimension
        self._width = resolution * (dimension - 1)  # base/voxelizer.py:28
        self._radii_type = radii_type
        self._density_type = density_type
        self.upper_bound = self._width / 2.0  # base/voxelizer.py:33-34
        self.lower_bound = -1 * self.upper_bound
        self._spatial_dimension = (dimension,) * 3
        if density_type == "gaussian":
            self._sigma = kwargs.get("sigma", 0.5)  # base/voxelizer.py:37-38

    # -- radii / density switches (base/voxelizer.py:40-78) ------------------------------------
    @property
    def radii_type(self) -> str:
        return self._radii_type

    @radii_type.setter
    def radii_type(self, value: str):
        assert value in self.RADII_TYPE_LIST
        self._radii_type = value

    is_radii_type_scalar = property(lambda self: self._radii_type == "scalar")
    is_radii_type_channel_wise = property(lambda self: self._radii_type == "channel-wise")
    is_radii_type_atom_wise = property(lambda self: self._radii_type == "atom-wise")

    @property
    def density_type(self) -> str:
        return self._density_type

    @density_type.setter
    def density_type(self, value: str):
        # A property setter receives one value, so switching back to gaussian resets sigma to
        # 0.5 exactly as in the reference (base/voxelizer.py:65-70).
        assert value in self.DENSITY_TYPE_LIST
        self._density_type = value
        if value == "gaussian":
            self._sigma = 0.5
        self._density_changed()

    def _density_changed(self):
        """Hook for backends holding native state."""

    is_density_type_binary = property(lambda self: self._density_type == "binary")
    is_density_type_gaussian = property(lambda self: self._density_type == "gaussian")

    # -- geometry (base/voxelizer.py:80-97) ---------------------------------------------------------
    def grid_dimension(self, num_channels: int):
        return (num_channels,) + self._spatial_dimension

    spatial_dimension = property(lambda self: self._spatial_dimension)
    resolution = property(lambda self: self._resolution)
    dimension = property(lambda self: self._dimension)
    width = property(lambda self: self._width)

    # -- forward dispatch (base/voxelizer.py:101-130) -----------------------------------------------
    def forward(self, coords, center, channels, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        """channels None -> forward_single; 1-D -> forward_types; otherwise forward_features."""
        if channels is None:
            return self.forward_single(coords, center, radii, random_translation, random_rotation, out_grid)
        if np.ndim(channels) == 1:
            return self.forward_types(coords, center, channels, radii, random_translation, random_rotation, out_grid)
        return self.forward_features(coords, center, channels, radii, random_translation, random_rotation, out_grid)

    __call__ = forward

    @abc.abstractmethod
    def forward_types(self, coords, center, types, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        ...

    @abc.abstractmethod
    def forward_features(self, coords, center, features, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        ...

    @abc.abstractmethod
    def forward_single(self, coords, center, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        ...

    @abc.abstractmethod
    def get_empty_grid(self, num_channels: int, batch_size=None, init_zero: bool = False):
        ...

    @abc.abstractmethod
    def asarray(self, array, obj: str):
        ...
