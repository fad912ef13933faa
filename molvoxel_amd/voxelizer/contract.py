"""The operator contract every backend object honours, and the grid geometry behind it.

What user code written against the reference relies on (`molvoxel/voxelizer/base/voxelizer.py:9-176`,
`base/transform.py:6-33`): the names of the attributes and methods, the three-way dispatch of `forward` on the
channel argument, the string-valued `radii_type` / `density_type` switches with their boolean views, and the grid
geometry derived from `(resolution, dimension)`. This module states that contract once:

* `GridGeometry`  - the numbers: width, bounds, shapes;
* `Choice`        - a validated string switch that also provides the `is_<name>_<value>` views;
* `VoxelizerContract` (exported as `BaseVoxelizer`) - geometry + switches + `forward` dispatch; a backend supplies
  `forward_features / forward_types / forward_single / get_empty_grid / asarray`;
* `TransformContract`, `RandomTransformContract` (exported as `BaseT`, `BaseRandomTransform`).
"""
from __future__ import annotations

import dataclasses

import numpy as np

RADII_TYPES = ("scalar", "channel-wise", "atom-wise")
DENSITY_TYPES = ("gaussian", "binary")
DEFAULT_SIGMA = 0.5


@dataclasses.dataclass(frozen=True)
class GridGeometry:
    """A cube of `dimension`^3 voxel centres, `resolution` apart, centred on the origin."""

    resolution: float
    dimension: int

    @property
    def width(self) -> float:
        return self.resolution * (self.dimension - 1)  # first to last voxel centre

    @property
    def upper_bound(self) -> float:
        return self.width / 2.0

    @property
    def lower_bound(self) -> float:
        return -1 * self.upper_bound

    @property
    def spatial_dimension(self) -> tuple:
        return (self.dimension,) * 3

    def grid_dimension(self, num_channels: int) -> tuple:
        return (num_channels,) + self.spatial_dimension


class Choice:
    """Class-level descriptor for a string switch restricted to `options`.

    `radii_type = Choice("radii_type", RADII_TYPES)` gives the instance a validated read/write attribute and the
    owner class read-only booleans `is_radii_type_scalar`, `is_radii_type_channel_wise`, ... (one per option,
    dashes spelt as underscores). `on_change` names an optional instance method called after a successful write.
    """

    def __init__(self, name: str, options, on_change: str | None = None):
        self.name, self.options, self.on_change = name, tuple(options), on_change
        self.slot = "_" + name

    def __set_name__(self, owner, attr):
        for option in self.options:
            view = f"is_{self.name}_{option.replace('-', '_')}"
            setattr(owner, view, property(lambda inst, _o=option, _s=self.slot: getattr(inst, _s) == _o))

    def __get__(self, inst, owner=None):
        return self if inst is None else getattr(inst, self.slot)

    def __set__(self, inst, value):
        assert value in self.options, f"{self.name} should be one of {list(self.options)}"
        setattr(inst, self.slot, value)
        if self.on_change is not None:
            getattr(inst, self.on_change)(value)


class RandomTransformContract:
    """Holds the two knobs; `forward(coords, center)` draws and applies, `get_transform()` freezes one draw."""

    class_T: type = None  # the backend's frozen-transform class

    def __init__(self, random_translation: float = 0.0, random_rotation: bool = False):
        self.random_translation = random_translation
        self.random_rotation = random_rotation

    def forward(self, coords, center):
        raise NotImplementedError

    def get_transform(self):
        return self.class_T.create(self.random_translation, self.random_rotation)


class TransformContract:
    """One drawn rigid transform: `T(coords, center) -> coords`; `T.create(translation, rotation)` draws it."""

    def __call__(self, coords, center):
        raise NotImplementedError

    @classmethod
    def create(cls, random_translation: float = 0.0, random_rotation: bool = False):
        raise NotImplementedError


class VoxelizerContract:
    LIB = None
    transform_class: type = RandomTransformContract
    RADII_TYPE_LIST = list(RADII_TYPES)
    DENSITY_TYPE_LIST = list(DENSITY_TYPES)

    radii_type = Choice("radii_type", RADII_TYPES)
    density_type = Choice("density_type", DENSITY_TYPES, on_change="_on_density_type")

    def __init__(self, resolution=0.5, dimension=48, radii_type="scalar", density_type="gaussian", **kwargs):
        self._geometry = GridGeometry(resolution, dimension)
        # the constructor honours `sigma=`; the setter (one value only) cannot, see _on_density_type
        self._radii_type = None
        self._density_type = None
        self.radii_type = radii_type
        assert density_type in DENSITY_TYPES
        self._density_type = density_type
        if density_type == "gaussian":
            self._sigma = kwargs.get("sigma", DEFAULT_SIGMA)

    # ---- switches ---------------------------------------------------------------------------------------
    def _on_density_type(self, value):
        """Assigning `density_type = "gaussian"` later falls back to the default sigma - a property setter receives
        one value, which is how the reference behaves too (`base/voxelizer.py:65-70`)."""
        if value == "gaussian":
            self._sigma = DEFAULT_SIGMA
        self._density_changed()

    def _density_changed(self):
        """For backends that keep native state in step."""

    # ---- geometry, under the names user code reads ---------------------------------------------------------
    _resolution = property(lambda self: self._geometry.resolution)
    _dimension = property(lambda self: self._geometry.dimension)
    resolution = property(lambda self: self._geometry.resolution)
    dimension = property(lambda self: self._geometry.dimension)
    width = property(lambda self: self._geometry.width)
    upper_bound = property(lambda self: self._geometry.upper_bound)
    lower_bound = property(lambda self: self._geometry.lower_bound)
    spatial_dimension = property(lambda self: self._geometry.spatial_dimension)

    def grid_dimension(self, num_channels: int) -> tuple:
        return self._geometry.grid_dimension(num_channels)

    # ---- the operator -----------------------------------------------------------------------------------
    def forward(self, coords, center, channels, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        """One entry point for the three operators, told apart by `channels`:
        None -> `forward_single`, a 1-D array of type indices -> `forward_types`, a 2-D array -> `forward_features`."""
        tail = (radii, random_translation, random_rotation, out_grid)
        if channels is None:
            return self.forward_single(coords, center, *tail)
        operator = self.forward_types if np.ndim(channels) == 1 else self.forward_features
        return operator(coords, center, channels, *tail)

    __call__ = forward

    def forward_types(self, coords, center, types, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        raise NotImplementedError

    def forward_features(self, coords, center, features, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        raise NotImplementedError

    def forward_single(self, coords, center, radii, random_translation=0.0, random_rotation=False, out_grid=None):
        raise NotImplementedError

    def get_empty_grid(self, num_channels: int, batch_size=None, init_zero: bool = False):
        raise NotImplementedError

    def asarray(self, array, obj: str):
        raise NotImplementedError


# the names the reference exports from molvoxel.voxelizer.base
BaseVoxelizer = VoxelizerContract
BaseRandomTransform = RandomTransformContract
BaseT = TransformContract
