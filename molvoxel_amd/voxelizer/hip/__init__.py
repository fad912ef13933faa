from .transform import RandomTransform, T
from .voxelizer import Voxelizer

__all__ = ["Voxelizer", "RandomTransform", "T"]
