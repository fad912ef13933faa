"""Backend factory, same call shape as molvoxel/voxelizer/__init__.py:9-37 of the reference.

Only one backend lives here: 'hip' (hand-written MI355X kernels behind a C ABI). The reference's
'numpy' / 'numba' / 'torch' backends are what it replaces; asking for them raises with a pointer to
the upstream package instead of silently running something else.
"""
from .base import BaseRandomTransform as RandomTransform
from .base import BaseVoxelizer as Voxelizer

LIBRARIES = ["hip"]


def _require_hip(library: str):
    assert library in LIBRARIES, (
        f"library={library!r} is not provided by molvoxel_amd (only {LIBRARIES}); "
        "the numpy/numba/torch backends belong to the upstream molvoxel package"
    )


def create_random_transform(random_translation: float = 0.0, random_rotation: bool = False, library: str = "hip", **kwargs) -> RandomTransform:
    _require_hip(library)
    from .hip import RandomTransform as TypeRandomTransform

    return TypeRandomTransform(random_translation, random_rotation, **kwargs)


def create_voxelizer(resolution: float = 0.5, dimension: int = 64, radii_type: str = "scalar", density_type: str = "gaussian",
                     library: str = "hip", **kwargs) -> Voxelizer:
    _require_hip(library)
    from .hip import Voxelizer as TypeVoxelizer

    return TypeVoxelizer(resolution, dimension, radii_type, density_type, **kwargs)
