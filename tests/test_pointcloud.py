"""Callers before the hot path: readers, getters, point-cloud makers, wrappers.

The reference's versions need RDKit (absent here), so expected values come from the reference's own test data
(tests/golden/10gs/*, copied data files) read by the independent parser of oracle/gen_golden.py
(tests/golden/pointcloud_10gs.npz) and from properties the reference's code documents (file:line in each test).
"""
import os

import numpy as np
import pytest

from molvoxel_amd.etc import mol as M

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PC = np.load(os.path.join(GOLD, "pointcloud_10gs.npz"))


@pytest.fixture(scope="module")
def ligand():
    return M.read_sdf(os.path.join(GOLD, "10gs", "10gs_ligand.sdf"))[0]


@pytest.fixture(scope="module")
def pocket():
    return M.read_pdb(os.path.join(GOLD, "10gs", "10gs_pocket_nowater.pdb"))


def test_readers_match_the_independent_parser(ligand, pocket):
    assert ligand.num_atoms == 33 and ligand.num_bonds == 34  # heavy atoms / bonds
    assert np.array_equal(ligand.coords, PC["ligand_xyz"])
    getter = M.AtomTypeGetter(["C", "N", "O", "S"], unknown=True)
    assert np.array_equal(getter.types_of_keys(ligand.symbols), PC["ligand_types"])
    assert np.array_equal(ligand.aromatic.astype(np.float32), PC["ligand_feat5"][:, 4])
    full = M.read_sdf(os.path.join(GOLD, "10gs", "10gs_ligand.sdf"), remove_hs=False)[0]
    assert full.num_atoms == 61 and full.num_bonds == 62
    assert int(full.charges[0]) == 1 and ligand.name == "10gs_ligand"
    # every pocket atom is one of the protein's heavy atoms
    prot = {tuple(r) for r in np.round(PC["protein_xyz"], 3).tolist()}
    assert pocket.num_atoms > 300 and all(tuple(r) in prot for r in np.round(pocket.coords, 3).tolist())
    assert "H" not in set(pocket.symbols)


def test_sdf_with_several_records_and_pdb_conect(tmp_path, ligand):
    src = open(os.path.join(GOLD, "10gs", "10gs_ligand.sdf")).read()
    two = tmp_path / "two.sdf"
    two.write_text(src + src)
    mols = M.read_sdf(str(two))
    assert len(mols) == 2 and all(np.array_equal(m.coords, ligand.coords) for m in mols)
    pdb = tmp_path / "w.pdb"
    pdb.write_text(
        "HETATM    1  O   HOH A   1       0.000   0.000   0.000  1.00  0.00           O  \n"
        "HETATM    2  H1  HOH A   1       0.957   0.000   0.000  1.00  0.00           H  \n"
        "HETATM    3 CL    CL A   2       3.000   0.000   0.000  1.00  0.00          CL  \n"
        "CONECT    1    2\nCONECT    1    3\nEND\n")
    m = M.read_pdb(str(pdb), remove_hs=False)
    assert list(m.symbols) == ["O", "H", "Cl"] and m.bonds.tolist() == [[0, 1], [0, 2]]
    assert M.read_pdb(str(pdb)).bonds.tolist() == [[0, 1]]  # hydrogen dropped, indices renumbered
    with pytest.raises(ValueError):
        bad = tmp_path / "v3000.mol"
        bad.write_text("x\n\n\n  0  0  0     0  0            999 V3000\nM  END\n")
        M.read_sdf(str(bad))


def test_getters_contract():
    """molvoxel/etc/rdkit/base.py:23-52, getter.py:14-46."""
    g = M.AtomTypeGetter(["C", "N"], unknown=True)
    assert g.channels == ["C", "N", "Unknown"] and g.num_channels == 3 and g.CHANNEL_TYPE == "TYPE"
    assert g.get_type("N") == 1 and g.get_type("Zn") == 2
    assert g.get_feature("C").tolist() == [1, 0, 0]
    strict = M.AtomTypeGetter(["C", "N"], ["carbon", "nitrogen"])
    assert strict.channels == ["carbon", "nitrogen"]
    with pytest.raises(KeyError):
        strict.get_type("O")
    fg = strict.to_feature_getter()
    assert fg.CHANNEL_TYPE == "FEATURE" and fg.channels == strict.channels and fg.get_feature("N").tolist() == [0, 1]
    b = M.BondTypeGetter.default()
    assert b.channels == ["SingleBond", "DoubleBond", "TripleBond", "AromaticBond"]
    assert [b.get_type(t) for t in (M.BondType.SINGLE, 2, M.BondType.AROMATIC)] == [0, 1, 3]
    assert M.BondTypeGetter([M.BondType.SINGLE, M.BondType.DOUBLE]).channels == ["SINGLE", "DOUBLE"]


def test_single_molecule_maker(ligand):
    """pointcloud.py:31-187: atoms then bond midpoints; types offset by the atom channels; one-hot features agree."""
    ag, bg = M.AtomTypeGetter(["C", "N", "O", "S"]), M.BondTypeGetter.default()
    mt = M.MolPointCloudMaker(ag, bg, channel_type="types")
    mf = M.MolPointCloudMaker(ag, bg, channel_type="features")
    assert mt.channels == ag.channels + bg.channels and mt.num_channels == 8
    coords, types = mt.run(ligand)
    assert coords.shape == (67, 3) and types.shape == (67,) and types.dtype == np.int16
    assert np.array_equal(coords[:33], ligand.coords)
    assert np.allclose(coords[33:], (ligand.coords[ligand.bonds[:, 0]] + ligand.coords[ligand.bonds[:, 1]]) / 2)
    assert types[:33].max() < 4 and types[33:].min() >= 4
    feats = mf.get_features(ligand)
    assert feats.dtype == np.float32 and feats.shape == (67, 8)
    assert np.array_equal(feats.argmax(axis=1), types) and np.array_equal(feats.sum(axis=1), np.ones(67))
    buf = np.full((67, 8), 7.0, dtype=np.float32)
    assert mf.get_channels(ligand, out=buf) is buf and np.array_equal(buf, feats)  # `out` is zeroed and refilled
    atoms_only = M.MolPointCloudMaker(ag, None, "types")
    assert atoms_only.get_coords(ligand).shape == (33, 3) and np.array_equal(atoms_only.get_types(ligand), types[:33])
    image = np.arange(8 * 2).reshape(8, 2)
    assert list(mt.split_channel(image)) == mt.channels and mt.split_channel(image)["O"].tolist() == [4, 5]
    with pytest.raises(AssertionError):
        M.MolPointCloudMaker(ag.to_feature_getter(), None, "types")  # a feature getter cannot give types (:46-48)
    with pytest.raises(AssertionError):
        M.MolPointCloudMaker(ag, None, "oops")


def test_feature_callbacks_see_rdkit_style_atoms(ligand):
    calls = []

    def fn(atom, scale=1.0):
        calls.append(atom.GetIdx())
        return [scale * (atom.GetSymbol() == "S"), float(atom.GetIsAromatic()), atom.GetFormalCharge()]

    maker = M.MolPointCloudMaker(M.AtomFeatureGetter(fn, ["isS", "arom", "charge"]), None, "features")
    f = maker.get_features(ligand, scale=2.0)
    assert calls == list(range(33)) and f.shape == (33, 3)
    assert f[:, 0].sum() == 2.0 * sum(s == "S" for s in ligand.symbols)
    assert np.array_equal(f[:, 1], PC["ligand_feat5"][:, 4]) and f[0, 2] == 1.0


def test_system_maker_channel_blocks(ligand, pocket):
    """pointcloud.py:211-326: each molecule gets its own block of channels, points are concatenated in order."""
    ag, bg = M.AtomTypeGetter(["C", "N", "O", "S"], unknown=True), M.BondTypeGetter.default()
    mk = M.ComplexPointCloudMaker(ag, bg, ag, None, channel_type="types")
    assert mk.num_channels == 5 + 4 + 5 and mk.channels[9:] == ag.channels
    coords, types = mk.run([ligand, pocket])
    nl = ligand.num_atoms + ligand.num_bonds
    assert coords.shape == (nl + pocket.num_atoms, 3)
    assert types[:33].max() <= 4 and 5 <= types[33:nl].min() and types[33:nl].max() <= 8 and types[nl:].min() >= 9
    assert np.array_equal(types[nl:] - 9, ag.types_of_keys(pocket.symbols))
    mkf = M.MolSystemPointCloudMaker(M.MolPointCloudMaker(ag, bg), (ag, None), channel_type="features")
    feats = mkf.get_channels([ligand, pocket])
    assert np.array_equal(feats.argmax(axis=1), types)
    parts = mk.split_channel(np.zeros((14, 2, 2, 2)))
    assert [len(p) for p in parts] == [9, 5]
    with pytest.raises(AssertionError):
        mk.run([ligand])


class _FakeRdAtom:
    def __init__(self, s):
        self.s = s

    def GetSymbol(self):
        return self.s

    def GetFormalCharge(self):
        return 0


class _FakeRdBond:
    def __init__(self, a, b, t):
        self.a, self.b, self.t = a, b, t

    def GetBeginAtomIdx(self):
        return self.a

    def GetEndAtomIdx(self):
        return self.b

    def GetBondType(self):
        return self.t


class _FakeRdMol:
    """Quacks like the parts of rdkit.Chem.Mol the reference's makers touch (pointcloud.py:78-85)."""

    def GetConformer(self):
        return self

    def GetPositions(self):
        return np.array([[0.0, 0, 0], [1.5, 0, 0], [3.0, 0, 0]])

    def GetAtoms(self):
        return [_FakeRdAtom(s) for s in "CNO"]

    def GetBonds(self):
        return [_FakeRdBond(0, 1, "AROMATIC"), _FakeRdBond(1, 2, "DOUBLE")]


def test_rdkit_objects_are_accepted_by_duck_typing():
    mk = M.MolPointCloudMaker(M.AtomTypeGetter(["C", "N", "O"]), M.BondTypeGetter.default(), "types")
    coords, types = mk.run(_FakeRdMol())
    assert coords[3:].tolist() == [[0.75, 0, 0], [2.25, 0, 0]] and types.tolist() == [0, 1, 2, 6, 4]
    with pytest.raises(TypeError):
        mk.run(object())


class _RecordingVoxelizer:
    """Stands in for a backend on the CPU: remembers what the wrapper passed to `forward` / `forward_batch`."""

    resolution, is_radii_type_scalar, is_radii_type_atom_wise, is_radii_type_channel_wise = 0.5, False, True, False

    def grid_dimension(self, c):
        return (c, 4, 4, 4)

    def asarray(self, a, obj):
        return np.asarray(a, dtype={"coords": np.float64, "center": np.float64, "types": np.int16}.get(obj, np.float32))

    def get_empty_grid(self, c, batch_size=None, init_zero=False):
        return np.zeros(((batch_size,) if batch_size else ()) + self.grid_dimension(c), np.float32)

    def forward(self, *args, **kw):
        self.call = (args, kw)
        return kw["out_grid"] if kw.get("out_grid") is not None else self.get_empty_grid(14)

    def forward_batch(self, *args, **kw):
        self.batch_call = (args, kw)
        return self.get_empty_grid(kw["num_channels"], len(args[1]) - 1)


def test_wrappers_pass_the_reference_arguments(ligand, pocket, tmp_path):
    """wrapper.py:21-45, 94-124: asarray conversions, per-molecule radii lists, out_grid shape check, same object back."""
    ag, bg = M.AtomTypeGetter(["C", "N", "O", "S"], unknown=True), M.BondTypeGetter.default()
    vox = _RecordingVoxelizer()
    w = M.ComplexWrapper(M.ComplexPointCloudMaker(ag, bg, ag, None, channel_type="types"), vox)
    assert w.num_channels == 14 and w.grid_dimension == (14, 4, 4, 4) and w.name_list == ["Ligand", "Protein"]
    grid = w.get_empty_grid()
    out = w.run(ligand, pocket, center=[1, 2, 3], radii=[1.0, 2.0], random_translation=0.5, random_rotation=True, out_grid=grid)
    (coords, center, channels, radii, tr, rot), kw = vox.call
    assert out is grid and kw["out_grid"] is grid and (tr, rot) == (0.5, True)
    assert coords.dtype == np.float64 and center.tolist() == [1, 2, 3] and channels.dtype == np.int16
    nl = 33 + 34
    assert radii.dtype == np.float32 and radii[:nl].tolist() == [1.0] * nl and set(radii[nl:].tolist()) == {2.0}
    with pytest.raises(AssertionError):
        w.run(ligand, pocket, out_grid=np.zeros((3, 4, 4, 4), np.float32))
    assert w.get_coords(ligand, pocket).shape == (nl + pocket.num_atoms, 3)
    # batch form: offsets per system, radii concatenated system by system
    w.run_batch([[ligand, pocket], [ligand, pocket]], centers=[[0, 0, 0], [1, 1, 1]], radii=[[1.0, 2.0], [3.0, 4.0]])
    (bcoords, offsets, centers, bch, bradii), bkw = vox.batch_call
    n = nl + pocket.num_atoms
    assert offsets.tolist() == [0, n, 2 * n] and bcoords.shape == (2 * n, 3) and centers.shape == (2, 3)
    assert bradii[n - 1] == 2.0 and bradii[n] == 3.0 and bkw["num_channels"] == 14
    paths = w.dump_dx(str(tmp_path), np.zeros((14, 4, 4, 4), np.float32), center=[0, 0, 0])
    assert len(paths) == 14 and os.path.basename(paths[0]) == "0_C.dx" and os.path.exists(paths[-1])


@pytest.mark.gpu
def test_wrapper_on_the_hip_backend_matches_the_oracle(ligand, pocket):
    import molvoxel_amd
    from oracle import c_oracle

    ag, bg = M.AtomTypeGetter(["C", "N", "O", "S"], unknown=True), M.BondTypeGetter.default()
    vox = molvoxel_amd.create_voxelizer(0.5, 48, "scalar", "gaussian", library="hip", output="numpy")
    center = ligand.coords.mean(axis=0)
    for channel_type in ("types", "features"):
        w = M.ComplexWrapper(M.ComplexPointCloudMaker(ag, bg, ag, None, channel_type=channel_type), vox)
        coords, channels = w.maker.run([ligand, pocket])
        img = w.run(ligand, pocket, center, radii=1.0)
        ref = c_oracle.voxelize(coords - center, channels, 1.0, dimension=48, density="gaussian", sigma=0.5, num_channels=14)
        # types without out_grid: max(types) + 1 channels (numpy/voxelizer.py:278); no pocket atom is "Unknown" here
        nc = 13 if channel_type == "types" else 14
        assert img.shape == (nc, 48, 48, 48) and not ref[13].any()
        assert np.array_equal(img != 0, ref[:nc] != 0) and np.abs(img - ref[:nc]).max() <= 5e-6
        batch = w.run_batch([[ligand, pocket]] * 3, centers=[center] * 3, radii=1.0)
        assert batch.shape == (3, 14, 48, 48, 48) and not batch[:, 13].any()
        assert all(np.array_equal(batch[b, :nc], img) for b in range(3))
