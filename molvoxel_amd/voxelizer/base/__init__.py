from .transform import BaseRandomTransform, BaseT
from .voxelizer import BaseVoxelizer

__all__ = ["BaseVoxelizer", "BaseRandomTransform", "BaseT"]
