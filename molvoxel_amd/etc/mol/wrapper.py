"""Wrappers: point-cloud maker + voxelizer = "molecule in, image out".

Same surface as the reference's `molvoxel/etc/rdkit/wrapper.py:11-217` (`MolWrapper`, `MolSystemWrapper`,
`ComplexWrapper`; `run`, `get_coords`, `get_channels`, `split_channel`, `get_empty_grid`), minus the PyMOL session
writer: `dump_dx` writes one OpenDX file per channel instead (`molvoxel_amd.etc.dx`).
Added for the GPU backend: `run_batch` voxelizes many molecules / systems in one launch (`Voxelizer.forward_batch`).
"""
from __future__ import annotations

import os

import numpy as np

from ..dx import write_grid_to_dx_file
from .molecule import as_molecule
from .pointcloud import ComplexPointCloudMaker, MolPointCloudMaker, MolSystemPointCloudMaker


class MolWrapper:
    def __init__(self, pointcloudmaker: MolPointCloudMaker, voxelizer):
        self.maker = pointcloudmaker
        self.voxelizer = voxelizer
        self.num_channels = self.maker.num_channels
        self.channel_type = self.maker.channel_type
        self.grid_dimension = self.voxelizer.grid_dimension(self.num_channels)
        self.resolution = self.voxelizer.resolution

    # -- the reference's entry points ---------------------------------------------------------------
    def run(self, mol, center=None, radii=1.0, random_translation: float = 0.0, random_rotation: bool = False,
            out_grid=None, **kwargs):
        coords, channels = self.maker.run(mol, **kwargs)
        return self._forward(coords, channels, center, self._radii(mol, radii), random_translation, random_rotation, out_grid)

    def get_coords(self, mol):
        return self.voxelizer.asarray(self.maker.get_coords(mol), "coords")

    def get_channels(self, mol):
        return self.voxelizer.asarray(self.maker.get_channels(mol), self.channel_type)

    def split_channel(self, image):
        return self.maker.split_channel(image)

    def get_empty_grid(self, batch_size: int | None = None, init_zero: bool = False):
        return self.voxelizer.get_empty_grid(self.num_channels, batch_size, init_zero)

    # -- shared plumbing ----------------------------------------------------------------------------
    def _radii(self, mol, radii):
        return radii

    def _forward(self, coords, channels, center, radii, random_translation, random_rotation, out_grid):
        vox = self.voxelizer
        if out_grid is not None:
            assert tuple(np.shape(out_grid)) == tuple(self.grid_dimension)
        coords = vox.asarray(coords, "coords")
        center = vox.asarray(center, "center") if center is not None else center
        channels = vox.asarray(channels, self.channel_type)
        radii = radii if np.isscalar(radii) else vox.asarray(radii, "radii")
        return vox.forward(coords, center, channels, radii, random_translation, random_rotation, out_grid=out_grid)

    def run_batch(self, mols, centers=None, radii=1.0, random_translation: float = 0.0, random_rotation: bool = False,
                  out_grid=None, **kwargs):
        """`run` for a list of molecules (or systems) in one launch; returns (B, C, D, H, W).
        `radii`: scalar, or one entry per molecule in the form `run` takes."""
        vox = self.voxelizer
        clouds = [self.maker.run(m, **kwargs) for m in mols]
        offsets = np.cumsum([0] + [c.shape[0] for c, _ in clouds]).astype(np.int64)
        coords = np.concatenate([c for c, _ in clouds], axis=0) if clouds else np.zeros((0, 3))
        channels = np.concatenate([ch for _, ch in clouds], axis=0)
        if not np.isscalar(radii):
            per_mol = [np.asarray(self._radii(m, r), dtype=np.float32) for m, r in zip(mols, radii)]
            radii = per_mol[0] if vox.is_radii_type_channel_wise else np.concatenate(per_mol)
            radii = vox.asarray(radii, "radii")
        if centers is not None:
            centers = np.asarray(centers, dtype=np.float64).reshape(len(clouds), 3)
        return vox.forward_batch(vox.asarray(coords, "coords"), offsets, centers, vox.asarray(channels, self.channel_type),
                                 radii, num_channels=self.num_channels, out_grid=out_grid,
                                 random_translation=random_translation, random_rotation=random_rotation)

    def dump_dx(self, directory: str, image, center=None, prefix: str = "") -> list[str]:
        """One `<prefix><channel>.dx` per channel (what the reference's visualizer feeds PyMOL, `etc/pymol/dx.py`)."""
        os.makedirs(directory, exist_ok=True)
        center = [0.0, 0.0, 0.0] if center is None else [float(v) for v in np.asarray(center).reshape(3)]
        named = self.split_channel(image)
        groups = named if isinstance(named, list) else [named]
        paths = []
        for g, group in enumerate(groups):
            for name, channel in group.items():
                tag = f"{prefix}{g}_" if len(groups) > 1 else prefix
                path = os.path.join(directory, f"{tag}{name}.dx")
                write_grid_to_dx_file(path, channel, center, self.resolution)
                paths.append(path)
        return paths


class MolSystemWrapper(MolWrapper):
    def __init__(self, pointcloudmaker: MolSystemPointCloudMaker, voxelizer, name_list: list[str] | None = None):
        super().__init__(pointcloudmaker, voxelizer)
        self.name_list = name_list

    def _radii(self, mol_list, radii):
        """A list gives one radius per molecule (atom-wise: repeated over its atoms) or the per-molecule channel
        radii to concatenate (channel-wise) - `wrapper.py:108-118`."""
        vox = self.voxelizer
        if vox.is_radii_type_scalar or not isinstance(radii, list):
            return radii
        if vox.is_radii_type_atom_wise:
            assert len(radii) == len(mol_list)
            counts = [section.num_points(as_molecule(m)) for m, section in zip(mol_list, self.maker.maker_list)]
            return np.concatenate([np.full(n, r, dtype=np.float32) for n, r in zip(counts, radii)])
        return np.concatenate([np.asarray(r, dtype=np.float32).reshape(-1) for r in radii])


class ComplexWrapper(MolSystemWrapper):
    def __init__(self, pointcloudmaker: ComplexPointCloudMaker, voxelizer):
        super().__init__(pointcloudmaker, voxelizer, ["Ligand", "Protein"])

    def run(self, ligand, protein, center=None, radii=1.0, random_translation: float = 0.0,
            random_rotation: bool = False, out_grid=None, **kwargs):
        return super().run([ligand, protein], center, radii, random_translation, random_rotation, out_grid, **kwargs)

    def get_coords(self, ligand, protein):
        return super().get_coords([ligand, protein])

    def get_channels(self, ligand, protein):
        return super().get_channels([ligand, protein])
