"""Import path kept for code that reads `molvoxel.voxelizer.base`; the contract itself is in `..contract`."""
from ..contract import BaseRandomTransform, BaseT, BaseVoxelizer

__all__ = ["BaseVoxelizer", "BaseRandomTransform", "BaseT"]
