"""Deterministic synthetic inputs for the five BASELINE.json configurations.

Used by bench.py, the tests and oracle/gen_golden.py so that every leg (reference import,
CPU oracle, HIP path) sees byte-identical inputs. Pure numpy; no GPU, no oracle.

Definitions follow SURVEY.md §8d / BASELINE.md §3:
  coords   = rng.uniform(-W/2, W/2, (N, 3)) float64, W = resolution * (dimension - 1)
  center   = zeros(3)
  features = rng.random((N, C)).astype(float32)
  cfg3 types = rng.integers(0, 4, N);  cfg5 radii = rng.uniform(1, 2, N).astype(float32)
  seeds: cfg2 -> 0, cfg3 -> 3, cfg4 -> 4, cfg5 -> 5; scalar radius 1.0 otherwise.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class Workload:
    name: str
    dimension: int
    resolution: float
    density: str
    sigma: float
    radii_type: str
    mode: str  # 'features' | 'types' | 'single'
    num_channels: int
    coords: list = field(default_factory=list)  # list of (N_i, 3) float64 (one entry per molecule)
    channels: list = field(default_factory=list)  # per molecule: (N_i, C) float32 | (N_i,) int | None
    radii: list = field(default_factory=list)  # per molecule: python float | (N_i,) float32
    centers: list = field(default_factory=list)  # per molecule: (3,) float64 (zeros for the synthetic configs)
    blockdim: int | None = None

    @property
    def batch(self) -> int:
        return len(self.coords)

    def algorithmic_bytes(self, i: int = 0) -> int:
        """every output voxel written once + one read of each atom record."""
        n = self.coords[i].shape[0]
        c = self.num_channels
        per_atom = 3 * 8 + 4
        if self.mode == "features":
            per_atom += 4 * c
        elif self.mode == "types":
            per_atom += 4
        return 4 * c * self.dimension**3 + n * per_atom


def _width(res, dim):
    return res * (dim - 1)


def cfg2(batch: int = 1, n_atoms: int = 4000, channels: int = 32, dimension: int = 64, seed: int = 0) -> Workload:
    """Protein pocket: N=4000, Gaussian sigma 0.5, 64^3, C=32, scalar radius 1.0 (the headline metric).
    Molecule 0 is the BASELINE seed-0 molecule; molecules i>0 use seed + 1000*i."""
    w = Workload("cfg2", dimension, 0.5, "gaussian", 0.5, "scalar", "features", channels)
    W = _width(0.5, dimension)
    for i in range(batch):
        rng = np.random.default_rng(seed + 1000 * i)
        w.coords.append(rng.uniform(-W / 2, W / 2, (n_atoms, 3)))
        w.channels.append(rng.random((n_atoms, channels)).astype(np.float32))
        w.radii.append(1.0)
        w.centers.append(np.zeros(3))
    return w


def cfg3(n_atoms: int = 1000, seed: int = 3, batch: int = 1) -> Workload:
    """Binary forward_types, 4 channels, 48^3, N=1000 (bit-exact check). Molecule 0 is the BASELINE seed-3 molecule;
    molecules i>0 use seed + 1000*i."""
    w = Workload("cfg3", 48, 0.5, "binary", 0.5, "scalar", "types", 4)
    W = _width(0.5, 48)
    for i in range(batch):
        rng = np.random.default_rng(seed + 1000 * i)
        w.coords.append(rng.uniform(-W / 2, W / 2, (n_atoms, 3)))
        w.channels.append(rng.integers(0, 4, n_atoms))
        w.radii.append(1.0)
        w.centers.append(np.zeros(3))
    return w


def ligand_like(rng, n_atoms: int) -> np.ndarray:
    """Compact chain of ~1.5 A steps, centred on the origin (a ligand-sized point cloud)."""
    steps = rng.normal(size=(n_atoms, 3))
    steps /= np.linalg.norm(steps, axis=1, keepdims=True)
    xyz = np.cumsum(1.5 * steps, axis=0)
    xyz *= 0.6  # chains fold: keep the radius of gyration ligand-like
    return xyz - xyz.mean(axis=0, keepdims=True)


def cfg4(batch: int = 1024, channels: int = 16, seed: int = 4) -> Workload:
    """Batch of ligands (N in [40, 60]), Gaussian, 64^3, C=16, scalar radius 1.0."""
    w = Workload("cfg4", 64, 0.5, "gaussian", 0.5, "scalar", "features", channels)
    rng = np.random.default_rng(seed)
    for _ in range(batch):
        n = int(rng.integers(40, 61))
        w.coords.append(ligand_like(rng, n))
        w.channels.append(rng.random((n, channels)).astype(np.float32))
        w.radii.append(1.0)
        w.centers.append(np.zeros(3))
    return w


def cfg5(n_atoms: int = 10000, seed: int = 5, batch: int = 1) -> Workload:
    """High-res 128^3, sigma 1.0, atom-wise radii in [1, 2), N=10000, C=32."""
    w = Workload("cfg5", 128, 0.5, "gaussian", 1.0, "atom-wise", "features", 32)
    W = _width(0.5, 128)
    for i in range(batch):
        rng = np.random.default_rng(seed + 1000 * i)
        w.coords.append(rng.uniform(-W / 2, W / 2, (n_atoms, 3)))
        w.channels.append(rng.random((n_atoms, 32)).astype(np.float32))
        w.radii.append(rng.uniform(1.0, 2.0, n_atoms).astype(np.float32))
        w.centers.append(np.zeros(3))
    return w


def cfg1(ligand_xyz: np.ndarray, ligand_feat: np.ndarray) -> Workload:
    """Single ligand (test/10gs fixture: 33 heavy atoms, C=5 = C,N,O,S,aromatic), Gaussian 0.5, 64^3."""
    w = Workload("cfg1", 64, 0.5, "gaussian", 0.5, "scalar", "features", ligand_feat.shape[1])
    xyz = np.asarray(ligand_xyz, dtype=np.float64)
    w.coords.append(xyz)
    w.channels.append(np.asarray(ligand_feat, dtype=np.float32))
    w.radii.append(1.0)
    w.centers.append(xyz.mean(axis=0))
    return w
