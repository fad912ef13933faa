"""Backend registry and the two factories user code calls.

`create_voxelizer(resolution, dimension, radii_type, density_type, library, **backend_kwargs)` and
`create_random_transform(random_translation, random_rotation, library, **backend_kwargs)` take the arguments of the
reference's factories (molvoxel/voxelizer/__init__.py:9-37). One backend is registered: 'hip', the MI355X kernels
behind the C ABI of include/mvx.h. The reference's 'numpy' / 'numba' / 'torch' backends are what it replaces; asking
for them fails with a pointer to the upstream package rather than running something else.
"""
import importlib

from .contract import BaseRandomTransform as RandomTransform
from .contract import BaseVoxelizer as Voxelizer

# library name -> module holding `Voxelizer` and `RandomTransform`
BACKENDS = {"hip": "molvoxel_amd.voxelizer.hip"}
LIBRARIES = list(BACKENDS)


def _backend(library: str):
    assert library in BACKENDS, (
        f"library={library!r} is not provided by molvoxel_amd (only {LIBRARIES}); "
        "the numpy/numba/torch backends belong to the upstream molvoxel package"
    )
    return importlib.import_module(BACKENDS[library])  # imported on demand: loading libmvx_hip.so needs no GPU


def create_voxelizer(resolution: float = 0.5, dimension: int = 64, radii_type: str = "scalar",
                     density_type: str = "gaussian", library: str = "hip", **kwargs) -> Voxelizer:
    """kwargs go to the backend: `sigma`, `blockdim`, `precision`, `device`, `output`."""
    return _backend(library).Voxelizer(resolution, dimension, radii_type, density_type, **kwargs)


def create_random_transform(random_translation: float = 0.0, random_rotation: bool = False, library: str = "hip",
                            **kwargs) -> RandomTransform:
    return _backend(library).RandomTransform(random_translation, random_rotation, **kwargs)
