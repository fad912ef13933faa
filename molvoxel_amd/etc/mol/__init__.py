"""Callers of the hot path without RDKit: molecule records and readers, channel getters, point-cloud makers,
wrappers (counterparts of the reference's `molvoxel/etc/rdkit/`)."""
from .getter import (AtomChannelGetter, AtomFeatureGetter, AtomTypeGetter, BondChannelGetter, BondFeatureGetter,
                     BondTypeGetter, ChannelGetter, FeatureGetter, TypeGetter)
from .molecule import BondType, Molecule, as_molecule, read_pdb, read_sdf
from .pointcloud import ComplexPointCloudMaker, MolPointCloudMaker, MolSystemPointCloudMaker, PointCloudMaker
from .wrapper import ComplexWrapper, MolSystemWrapper, MolWrapper

__all__ = [
    "AtomChannelGetter", "AtomFeatureGetter", "AtomTypeGetter", "BondChannelGetter", "BondFeatureGetter",
    "BondTypeGetter", "ChannelGetter", "FeatureGetter", "TypeGetter", "BondType", "Molecule", "as_molecule",
    "read_pdb", "read_sdf", "ComplexPointCloudMaker", "MolPointCloudMaker", "MolSystemPointCloudMaker",
    "PointCloudMaker", "ComplexWrapper", "MolSystemWrapper", "MolWrapper",
]
