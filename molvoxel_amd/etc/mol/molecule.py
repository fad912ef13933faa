"""A plain-array molecule record and RDKit-free readers for the files the reference's tests use.

The reference builds its point clouds from RDKit `Mol` objects (`molvoxel/etc/rdkit/pointcloud.py:78-88`:
`GetConformer().GetPositions()`, `GetAtoms()`, `GetBonds()`). RDKit is not part of this stack, so the callers of
the hot path work on `Molecule`: coordinates, element symbols and bonds as arrays. An RDKit `Mol` (when the user
has RDKit) converts with `Molecule.from_rdkit` and keeps RDKit's own perception (aromaticity, removed hydrogens).

Readers: MDL molfile/SDF V2000 (`read_sdf`) and PDB `ATOM`/`HETATM` records (`read_pdb`). Like RDKit's defaults
(`SDMolSupplier(removeHs=True)`, `MolFromPDBFile(removeHs=True)`) hydrogens are dropped unless asked for. Bond orders
are taken as written (1, 2, 3, 4 = aromatic); nothing is perceived or sanitised.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field

import numpy as np


class BondType(enum.IntEnum):
    """Bond orders as the molfile writes them (the names RDKit's `BondType` uses)."""

    SINGLE = 1
    DOUBLE = 2
    TRIPLE = 3
    AROMATIC = 4

    def __str__(self):
        return self.name


# molfile V2000 atom-block charge codes (columns 37-39)
_SDF_CHARGE = {0: 0, 1: 3, 2: 2, 3: 1, 4: 0, 5: -1, 6: -2, 7: -3}


@dataclass
class Molecule:
    """coords (N,3) float64; symbols (N,) str; bonds (M,2) int64 atom indices; bond_types (M,) int (BondType)."""

    coords: np.ndarray
    symbols: np.ndarray
    bonds: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.int64))
    bond_types: np.ndarray = field(default_factory=lambda: np.zeros((0,), dtype=np.int64))
    charges: np.ndarray | None = None
    name: str = ""

    def __post_init__(self):
        self.coords = np.ascontiguousarray(self.coords, dtype=np.float64).reshape(-1, 3)
        self.symbols = np.asarray(self.symbols, dtype=object)
        self.bonds = np.asarray(self.bonds, dtype=np.int64).reshape(-1, 2)
        self.bond_types = np.asarray(self.bond_types, dtype=np.int64).reshape(-1)
        if self.charges is None:
            self.charges = np.zeros(self.num_atoms, dtype=np.int64)
        assert self.symbols.shape == (self.num_atoms,), "one symbol per atom"
        assert self.bond_types.shape == (self.num_bonds,), "one bond type per bond"
        assert self.num_bonds == 0 or (self.bonds.min() >= 0 and self.bonds.max() < self.num_atoms), "bond index out of range"

    # -- the few queries the point-cloud makers need ------------------------------------------------
    @property
    def num_atoms(self) -> int:
        return int(self.coords.shape[0])

    @property
    def num_bonds(self) -> int:
        return int(self.bonds.shape[0])

    @property
    def aromatic(self) -> np.ndarray:
        """(N,) bool: atom is an end of an aromatic bond."""
        flag = np.zeros(self.num_atoms, dtype=bool)
        arom = self.bonds[self.bond_types == BondType.AROMATIC]
        flag[arom.reshape(-1)] = True
        return flag

    def bond_centers(self) -> np.ndarray:
        """(M,3): midpoints, the positions the reference gives its bond points (`pointcloud.py:82-85`)."""
        return (self.coords[self.bonds[:, 0]] + self.coords[self.bonds[:, 1]]) / 2

    def without_hydrogens(self) -> "Molecule":
        keep = np.array([s not in ("H", "D") for s in self.symbols], dtype=bool)
        if keep.all():
            return self
        new_index = np.cumsum(keep) - 1
        bond_keep = keep[self.bonds].all(axis=1) if self.num_bonds else np.zeros(0, dtype=bool)
        return Molecule(self.coords[keep], self.symbols[keep], new_index[self.bonds[bond_keep]],
                        self.bond_types[bond_keep], self.charges[keep], self.name)

    # -- RDKit-shaped views, so getter callbacks written for RDKit atoms/bonds keep working ----------
    def atoms(self):
        arom = self.aromatic
        return [AtomView(self, i, bool(arom[i])) for i in range(self.num_atoms)]

    def bond_views(self):
        return [BondView(self, j) for j in range(self.num_bonds)]

    @classmethod
    def from_rdkit(cls, rdmol) -> "Molecule":
        """Convert an RDKit `Mol` (duck-typed: nothing is imported)."""
        coords = np.asarray(rdmol.GetConformer().GetPositions(), dtype=np.float64)
        symbols = [a.GetSymbol() for a in rdmol.GetAtoms()]
        charges = [a.GetFormalCharge() for a in rdmol.GetAtoms()]
        order = {"SINGLE": 1, "DOUBLE": 2, "TRIPLE": 3, "AROMATIC": 4}
        bonds = [(b.GetBeginAtomIdx(), b.GetEndAtomIdx()) for b in rdmol.GetBonds()]
        types = [order.get(str(b.GetBondType()), 0) for b in rdmol.GetBonds()]
        return cls(coords, symbols, np.array(bonds, dtype=np.int64).reshape(-1, 2), types, np.array(charges))


class AtomView:
    """One atom with the accessor names RDKit's `Atom` has (what `AtomTypeGetter`/feature callbacks call)."""

    __slots__ = ("mol", "idx", "_aromatic")

    def __init__(self, mol: Molecule, idx: int, aromatic: bool):
        self.mol, self.idx, self._aromatic = mol, idx, aromatic

    def GetSymbol(self) -> str:
        return str(self.mol.symbols[self.idx])

    def GetIdx(self) -> int:
        return self.idx

    def GetIsAromatic(self) -> bool:
        return self._aromatic

    def GetFormalCharge(self) -> int:
        return int(self.mol.charges[self.idx])


class BondView:
    __slots__ = ("mol", "idx")

    def __init__(self, mol: Molecule, idx: int):
        self.mol, self.idx = mol, idx

    def GetBondType(self) -> BondType:
        return BondType(int(self.mol.bond_types[self.idx]))

    def GetBeginAtomIdx(self) -> int:
        return int(self.mol.bonds[self.idx, 0])

    def GetEndAtomIdx(self) -> int:
        return int(self.mol.bonds[self.idx, 1])

    def GetIsAromatic(self) -> bool:
        return int(self.mol.bond_types[self.idx]) == BondType.AROMATIC


def as_molecule(mol) -> Molecule:
    """`Molecule` as is; anything with `GetConformer` (an RDKit Mol) is converted."""
    if isinstance(mol, Molecule):
        return mol
    if hasattr(mol, "GetConformer"):
        return Molecule.from_rdkit(mol)
    raise TypeError(f"expected a Molecule or an RDKit Mol, got {type(mol).__name__}")


# ---------------------------------------------------------------------------------------------------
# readers
# ---------------------------------------------------------------------------------------------------
def _parse_molblock(lines: list[str], remove_hs: bool) -> Molecule:
    if len(lines) < 4 or "V2000" not in lines[3]:
        raise ValueError("only V2000 molfiles are supported")
    counts = lines[3]
    na, nb = int(counts[0:3]), int(counts[3:6])
    if len(lines) < 4 + na + nb:
        raise ValueError("truncated molfile")
    coords = np.empty((na, 3), dtype=np.float64)
    symbols, charges = [], np.zeros(na, dtype=np.int64)
    for i, ln in enumerate(lines[4 : 4 + na]):
        coords[i] = (float(ln[0:10]), float(ln[10:20]), float(ln[20:30]))
        symbols.append(ln[31:34].strip())
        code = ln[36:39].strip()
        charges[i] = _SDF_CHARGE.get(int(code), 0) if code else 0
    bonds = np.empty((nb, 2), dtype=np.int64)
    types = np.empty(nb, dtype=np.int64)
    for j, ln in enumerate(lines[4 + na : 4 + na + nb]):
        bonds[j] = (int(ln[0:3]) - 1, int(ln[3:6]) - 1)
        types[j] = int(ln[6:9])
    for ln in lines[4 + na + nb :]:  # property block: M  CHG overrides the atom-block codes
        if ln.startswith("M  CHG"):
            n = int(ln[6:9])
            vals = ln[9:].split()
            for k in range(n):
                charges[int(vals[2 * k]) - 1] = int(vals[2 * k + 1])
        elif ln.startswith("M  END"):
            break
    mol = Molecule(coords, symbols, bonds, types, charges, lines[0].strip())
    return mol.without_hydrogens() if remove_hs else mol


def read_sdf(path: str, remove_hs: bool = True) -> list[Molecule]:
    """All records of an SDF file (a bare molfile gives one). A record is a molblock (three header lines, counts
    line, atoms, bonds, properties) up to the `$$$$` line."""
    with open(path) as fh:
        lines = fh.read().split("\n")
    mols, cur = [], []
    for ln in lines:
        if ln.startswith("$$$$"):
            mols.append(_parse_molblock(cur, remove_hs))
            cur = []
        else:
            cur.append(ln)
    if len(cur) >= 4 and any(ln.strip() for ln in cur):  # a bare molfile, or a last record without its $$$$
        mols.append(_parse_molblock(cur, remove_hs))
    return mols


def read_pdb(path: str, remove_hs: bool = True, hetatm: bool = True) -> Molecule:
    """`ATOM` (and `HETATM`) records: columns 31-54 coordinates, 77-78 element (falls back to the atom name);
    `CONECT` records become single bonds. No proximity bonding, no bond-order assignment."""
    coords, symbols, serials = [], [], {}
    conect = []
    with open(path) as fh:
        for ln in fh:
            rec = ln[0:6]
            if rec == "ATOM  " or (hetatm and rec == "HETATM"):
                el = ln[76:78].strip() if len(ln) >= 78 else ""
                if not el:
                    el = "".join(ch for ch in ln[12:16] if ch.isalpha())[:1]
                el = el.capitalize()
                serials[ln[6:11].strip()] = len(coords)
                coords.append((float(ln[30:38]), float(ln[38:46]), float(ln[46:54])))
                symbols.append(el)
            elif rec == "CONECT":
                conect.append(ln)
            elif rec.startswith("ENDMDL"):
                break
    pairs = set()
    for ln in conect:
        ids = [ln[6 + 5 * k : 11 + 5 * k].strip() for k in range(5)]
        ids = [serials[i] for i in ids if i and i in serials]
        for other in ids[1:]:
            if ids and other != ids[0]:
                pairs.add((min(ids[0], other), max(ids[0], other)))
    bonds = np.array(sorted(pairs), dtype=np.int64).reshape(-1, 2)
    mol = Molecule(np.array(coords, dtype=np.float64).reshape(-1, 3), symbols, bonds, np.ones(len(bonds), dtype=np.int64))
    return mol.without_hydrogens() if remove_hs else mol
