"""Point-cloud makers: molecules -> (coords, channels), the two arrays the voxelizer takes.

The caller step before the hot path, with the public surface of the reference's
`molvoxel/etc/rdkit/pointcloud.py:12-326` (`PointCloudMaker`, `MolPointCloudMaker`, `MolSystemPointCloudMaker`,
`ComplexPointCloudMaker`; `run`, `get_coords`, `get_channels`, `get_features`, `get_types`, `split_channel`,
`channels`, `num_channels`, `channel_type`). Points are the atoms followed by the bond midpoints (when a bond getter is
given); channels are laid out molecule by molecule, atoms before bonds.

Built differently from the reference: a system is a list of `_Section`s (getter pair + first channel), and type
getters are evaluated on whole symbol / bond-order arrays instead of per RDKit object.
"""
from __future__ import annotations

import numpy as np

from .getter import ChannelGetter, TypeGetter
from .molecule import Molecule, as_molecule


class PointCloudMaker:
    def __init__(self, channels: list[str]):
        self.channels = list(channels)
        self.num_channels = len(self.channels)

    def split_channel(self, image) -> dict:
        assert np.shape(image)[0] == self.num_channels
        return {name: image[i] for i, name in enumerate(self.channels)}

    def run(self, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.run(*args, **kwargs)


def _check_channel_type(channel_type: str):
    assert channel_type in ["features", "types"], f"channel_type(input: {channel_type}) should be 'features' or 'types'"


class _Section:
    """One molecule's share of the points and channels: atoms [start, start + A), bonds [start + A, start + A + B)."""

    def __init__(self, atom_getter: ChannelGetter, bond_getter: ChannelGetter | None, channel_type: str, start: int):
        allowed = ["TYPE", "FEATURE"] if channel_type == "features" else ["TYPE"]
        assert atom_getter.CHANNEL_TYPE in allowed
        assert bond_getter is None or bond_getter.CHANNEL_TYPE in allowed
        self.atom_getter, self.bond_getter = atom_getter, bond_getter
        self.use_bond = bond_getter is not None
        self.atom_start = start
        self.bond_start = start + atom_getter.num_channels
        self.num_atom_channels = atom_getter.num_channels
        self.num_bond_channels = bond_getter.num_channels if self.use_bond else 0
        self.num_channels = self.num_atom_channels + self.num_bond_channels
        self.channels = atom_getter.channels + (bond_getter.channels if self.use_bond else [])

    def num_points(self, mol: Molecule) -> int:
        return mol.num_atoms + (mol.num_bonds if self.use_bond else 0)

    def coords(self, mol: Molecule) -> np.ndarray:
        return np.concatenate([mol.coords, mol.bond_centers()], axis=0) if self.use_bond else mol.coords

    @staticmethod
    def _types(getter, keys, items, **kwargs) -> np.ndarray:
        if isinstance(getter, TypeGetter) and not kwargs:
            return getter.types_of_keys(keys)
        return np.fromiter((getter.get_type(it, **kwargs) for it in items()), dtype=np.int16, count=len(keys))

    def fill_types(self, mol: Molecule, out: np.ndarray, **kwargs):
        na = mol.num_atoms
        out[:na] = self._types(self.atom_getter, mol.symbols, mol.atoms, **kwargs) + self.atom_start
        if self.use_bond:
            out[na:] = self._types(self.bond_getter, mol.bond_types.tolist(), mol.bond_views, **kwargs) + self.bond_start

    @staticmethod
    def _features(getter, keys, items, out_block: np.ndarray, **kwargs):
        if len(keys) == 0:
            return
        if isinstance(getter, TypeGetter) and not kwargs:
            out_block[np.arange(len(keys)), getter.types_of_keys(keys)] = 1.0
        else:
            out_block[:] = [getter.get_feature(it, **kwargs) for it in items()]

    def fill_features(self, mol: Molecule, out: np.ndarray, **kwargs):
        """`out`: this molecule's rows of the (points, all channels) matrix, already zeroed."""
        na = mol.num_atoms
        self._features(self.atom_getter, mol.symbols, mol.atoms,
                       out[:na, self.atom_start : self.atom_start + self.num_atom_channels], **kwargs)
        if self.use_bond:
            self._features(self.bond_getter, mol.bond_types.tolist(), mol.bond_views,
                           out[na:, self.bond_start : self.bond_start + self.num_bond_channels], **kwargs)


class MolSystemPointCloudMaker(PointCloudMaker):
    """Several molecules voxelized into one image, each with its own block of channels.
    Arguments: `(atom_getter, bond_getter | None)` pairs or `MolPointCloudMaker`s, one per molecule."""

    def __init__(self, *args, channel_type: str = "features"):
        _check_channel_type(channel_type)
        self.channel_type = channel_type
        self.use_features = channel_type == "features"
        self.maker_list: list[_Section] = []
        start = 0
        for arg in args:
            atom_getter, bond_getter = (arg.atom_getter, arg.bond_getter) if isinstance(arg, MolPointCloudMaker) else arg
            section = _Section(atom_getter, bond_getter, channel_type, start)
            self.maker_list.append(section)
            start += section.num_channels
        super().__init__([name for section in self.maker_list for name in section.channels])

    def _pairs(self, mol_list):
        mols = [as_molecule(m) for m in mol_list]
        assert len(mols) == len(self.maker_list), f"expected {len(self.maker_list)} molecules, got {len(mols)}"
        return list(zip(mols, self.maker_list))

    def run(self, mol_list, **kwargs):
        return self.get_coords(mol_list), self.get_channels(mol_list, **kwargs)

    def get_coords(self, mol_list) -> np.ndarray:
        return np.concatenate([section.coords(mol) for mol, section in self._pairs(mol_list)], axis=0)

    def get_channels(self, mol_list, out=None, **kwargs):
        return self.get_features(mol_list, out, **kwargs) if self.use_features else self.get_types(mol_list, out, **kwargs)

    def get_features(self, mol_list, out=None, **kwargs) -> np.ndarray:
        pairs = self._pairs(mol_list)
        total = sum(section.num_points(mol) for mol, section in pairs)
        if out is None:
            out = np.zeros((total, self.num_channels), dtype=np.float32)
        else:
            assert out.shape == (total, self.num_channels)
            out.fill(0)
        row = 0
        for mol, section in pairs:
            n = section.num_points(mol)
            section.fill_features(mol, out[row : row + n], **kwargs)
            row += n
        return out

    def get_types(self, mol_list, out=None, **kwargs) -> np.ndarray:
        assert self.use_features is False
        pairs = self._pairs(mol_list)
        total = sum(section.num_points(mol) for mol, section in pairs)
        if out is None:
            out = np.empty((total,), dtype=np.int16)
        else:
            assert out.shape == (total,)
        row = 0
        for mol, section in pairs:
            n = section.num_points(mol)
            section.fill_types(mol, out[row : row + n], **kwargs)
            row += n
        return out

    def split_channel(self, image) -> list[dict]:
        assert np.shape(image)[0] == self.num_channels
        out, start = [], 0
        for section in self.maker_list:
            out.append({name: image[start + i] for i, name in enumerate(section.channels)})
            start += section.num_channels
        return out


class MolPointCloudMaker(PointCloudMaker):
    """One molecule (`pointcloud.py:31-74`): a one-section system behind the single-molecule signatures."""

    def __init__(self, atom_getter: ChannelGetter, bond_getter: ChannelGetter | None = None, channel_type: str = "features"):
        _check_channel_type(channel_type)
        self.channel_type = channel_type
        self.use_features = channel_type == "features"
        self.atom_getter, self.bond_getter = atom_getter, bond_getter
        self.use_bond = bond_getter is not None
        self._system = MolSystemPointCloudMaker((atom_getter, bond_getter), channel_type=channel_type)
        self.num_atom_channels = atom_getter.num_channels
        if self.use_bond:
            self.num_bond_channels = bond_getter.num_channels
        super().__init__(self._system.channels)

    def run(self, mol, **kwargs):
        return self._system.run([mol], **kwargs)

    def get_coords(self, mol) -> np.ndarray:
        return self._system.get_coords([mol])

    def get_channels(self, mol, out=None, **kwargs):
        return self._system.get_channels([mol], out, **kwargs)

    def get_features(self, mol, out=None, **kwargs):
        return self._system.get_features([mol], out, **kwargs)

    def get_types(self, mol, out=None, **kwargs):
        return self._system.get_types([mol], out, **kwargs)


class ComplexPointCloudMaker(MolSystemPointCloudMaker):
    """Ligand + protein (`pointcloud.py:311-326`)."""

    def __init__(self, ligand_atom_getter, ligand_bond_getter, protein_atom_getter, protein_bond_getter,
                 channel_type: str = "features"):
        super().__init__((ligand_atom_getter, ligand_bond_getter), (protein_atom_getter, protein_bond_getter),
                         channel_type=channel_type)
