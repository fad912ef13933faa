"""molvoxel_amd — MI355X-native drop-in for molvoxel's voxelizer hot path (library='hip').

    from molvoxel_amd import create_voxelizer
    vox = create_voxelizer(0.5, 64, 'scalar', 'gaussian', library='hip')
    grid = vox.forward_features(coords, center, features, 1.0)      # torch CUDA tensor (C, 64, 64, 64)

Same factory signature and operator contract as the reference (molvoxel/__init__.py:9-40).
Importing this package does not need a GPU; creating a voxelizer does (no CPU fallback).
"""
from . import voxelizer
from .voxelizer import create_random_transform, create_voxelizer
from .voxelizer.contract import BaseRandomTransform as RandomTransform
from .voxelizer.contract import BaseVoxelizer as Voxelizer

__version__ = "0.1.0"

__all__ = ["create_voxelizer", "create_random_transform", "Voxelizer", "RandomTransform", "voxelizer", "__version__"]
