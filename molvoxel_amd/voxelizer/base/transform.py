"""Abstract transform contract (mirrors molvoxel/voxelizer/base/transform.py:6-33 of the reference)."""
from __future__ import annotations

import abc


class BaseT(abc.ABC):
    """A frozen rigid transform: T(coords, center) -> coords."""

    @abc.abstractmethod
    def __call__(self, coords, center):
        ...

    @classmethod
    @abc.abstractmethod
    def create(cls, random_translation: float = 0.0, random_rotation: bool = False):
        ...


class BaseRandomTransform(abc.ABC):
    class_T = BaseT

    def __init__(self, random_translation: float = 0.0, random_rotation: bool = False):
        self.random_translation = random_translation
        self.random_rotation = random_rotation

    @abc.abstractmethod
    def forward(self, coords, center):
        ...

    def get_transform(self) -> BaseT:
        return self.class_T.create(self.random_translation, self.random_rotation)
