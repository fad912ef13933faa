"""Quaternion helpers for the host side of the HIP backend.

RNG-order and arithmetic-order compatible with the reference (molvoxel/voxelizer/numpy/_quaternion.py:13-50):
a uniform unit quaternion from three np.random.rand draws, rotation as q * (0, p) * q^-1 evaluated
with the same products and the same left-to-right sums, so seeded results are identical. The device
applies the same expression tree in fp64 (csrc/mvx_device.h: apply_xform).
"""
from __future__ import annotations

import math

import numpy as np

TWO_PI = 2 * math.pi


def random_quaternion():
    """Draws exactly three uniforms from the global numpy RNG (_quaternion.py:13-21)."""
    u1, u2, u3 = np.random.rand(3)
    a, b = math.sqrt(1 - u1), math.sqrt(u1)
    return (a * math.sin(TWO_PI * u2), a * math.cos(TWO_PI * u2), b * math.sin(TWO_PI * u3), b * math.cos(TWO_PI * u3))


def _hamilton(p, q):
    p0, p1, p2, p3 = p
    q0, q1, q2, q3 = q
    return (
        p0 * q0 - p1 * q1 - p2 * q2 - p3 * q3,
        p0 * q1 + p1 * q0 + p2 * q3 - p3 * q2,
        p0 * q2 - p1 * q3 + p2 * q0 + p3 * q1,
        p0 * q3 + p1 * q2 - p2 * q1 + p3 * q0,
    )


def rotate(points: np.ndarray, quaternion) -> np.ndarray:
    """(N, 3) -> (N, 3): q * (0, p) * conj(q) (_quaternion.py:45-50)."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    conj = (quaternion[0], quaternion[1] * -1, quaternion[2] * -1, quaternion[3] * -1)
    _, rx, ry, rz = _hamilton(_hamilton(quaternion, (np.zeros_like(x), x, y, z)), conj)
    return np.stack([rx, ry, rz], axis=-1)
