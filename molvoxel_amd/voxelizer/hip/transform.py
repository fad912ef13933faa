"""Random rigid transforms for the HIP backend.

Host objects with the reference's interface and RNG behaviour
(molvoxel/voxelizer/numpy/transform.py:11-80): `T.create` draws the translation first and then the
quaternion; `do_random_transform` (what forward_* uses) draws the quaternion first and then the
translation; with both a rotation and a translation the translation is applied twice (reference
quirk) — kept for seeded parity. numpy inputs are transformed with numpy; torch CUDA
tensors are transformed on the device by mvx_transform_coords.
"""
from __future__ import annotations

import numpy as np

from ..contract import BaseRandomTransform, BaseT
from ._quaternion import random_quaternion, rotate


def draw_translation(random_translation: float):
    return np.random.uniform(-random_translation, random_translation, size=(1, 3)).astype(np.float32)


def draw_forward_transform(random_translation, random_rotation):
    """RNG order used inside forward_* (numpy/transform.py:63-80): quaternion, then translation."""
    quaternion = random_quaternion() if random_rotation else None
    translation = None
    if random_translation is not None and random_translation > 0.0:
        translation = draw_translation(random_translation)
    return translation, quaternion


def _is_torch_cuda(x) -> bool:
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def do_transform(coords, center=None, translation=None, quaternion=None):
    if _is_torch_cuda(coords):
        from .voxelizer import transform_on_device

        return transform_on_device(coords, center, translation, quaternion)
    coords = np.asarray(coords)
    if quaternion is not None:
        if center is not None:
            c = np.asarray(center).reshape(1, 3)
            coords = rotate(coords - c, quaternion)
            coords += c
        else:
            coords = rotate(coords, quaternion)
        if translation is not None:
            coords += translation
    if translation is not None:
        coords = coords + translation
    return coords


def do_random_transform(coords, center=None, random_translation=0.0, random_rotation=False):
    translation, quaternion = draw_forward_transform(random_translation, random_rotation)
    return do_transform(coords, center, translation, quaternion)


class T(BaseT):
    def __init__(self, translation, quaternion):
        self.translation = translation
        self.quaternion = quaternion

    def __call__(self, coords, center):
        return do_transform(coords, center, self.translation, self.quaternion)

    @classmethod
    def create(cls, random_translation: float = 0.0, random_rotation: bool = False):
        translation = draw_translation(random_translation) if random_translation > 0.0 else None
        quaternion = random_quaternion() if random_rotation else None
        return cls(translation, quaternion)


class RandomTransform(BaseRandomTransform):
    class_T = T

    def forward(self, coords, center):
        return do_random_transform(coords, center, self.random_translation, self.random_rotation)
