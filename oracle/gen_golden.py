#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE numpy backend (imported from /root/reference).

TEST INFRASTRUCTURE ONLY. Runs in the build container only (the reference never travels to
the GPU box); the fixtures it writes are data: inputs + the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]

Writes:
  tests/golden/small_cases.npz   full output arrays, small grids, every mode x radii x density,
                                 incl. atoms outside the box, exact-tie cases, res 0.4, odd blockdim
  tests/golden/big_cases.npz     BASELINE configs 1/2/3/4(8 ligands)/5: sha256 (binary / integer-valued
                                 outputs), per-channel float64 sums, 10^4 sampled voxel values
  tests/golden/pointcloud_10gs.npz   heavy-atom point clouds parsed from the reference's test/10gs data
  tests/golden/transform_cases.npz   seeded random-transform cases (np.random.seed -> coords out)
  tests/golden/api_cases.npz     end-to-end Voxelizer.forward calls through the reference's public API
  tests/golden/dx_cases.npz      OpenDX dumps written by the reference's dx.py
  tests/golden/dense_cases.npz   dense clusters (sums of hundreds of terms): full arrays
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


# ----------------------------------------------------------------------------- 10gs readers
def read_sdf_heavy(path):
    """V2000 molfile: heavy-atom coords, element symbols, aromatic flag (any bond of type 4)."""
    with open(path) as fh:
        lines = fh.read().splitlines()
    na, nbonds = int(lines[3][0:3]), int(lines[3][3:6])
    xyz, sym = [], []
    for ln in lines[4 : 4 + na]:
        xyz.append([float(ln[0:10]), float(ln[10:20]), float(ln[20:30])])
        sym.append(ln[31:34].strip())
    arom = np.zeros(na, dtype=bool)
    for ln in lines[4 + na : 4 + na + nbonds]:
        a, b, t = int(ln[0:3]) - 1, int(ln[3:6]) - 1, int(ln[6:9])
        if t == 4:
            arom[a] = arom[b] = True
    heavy = np.array([s != "H" for s in sym])
    return np.array(xyz)[heavy], [s for s, h in zip(sym, heavy) if h], arom[heavy]


def read_pdb_heavy(path):
    xyz, sym = [], []
    with open(path) as fh:
        for ln in fh:
            if ln.startswith(("ATOM", "HETATM")):
                el = ln[76:78].strip().capitalize()
                if el == "H":
                    continue
                xyz.append([float(ln[30:38]), float(ln[38:46]), float(ln[46:54])])
                sym.append(el)
    return np.array(xyz), sym


def element_types(sym, table=("C", "N", "O", "S")):
    """Index into `table`; unknown elements get len(table)."""
    return np.array([table.index(s) if s in table else len(table) for s in sym], dtype=np.int16)


# ----------------------------------------------------------------------------- small cases
def small_inputs(rng, D, res, n_atoms, n_chan):
    W = res * (D - 1)
    xyz = rng.uniform(-W / 2 - 1.5, W / 2 + 1.5, (n_atoms, 3))  # some atoms outside the box
    axis = np.arange(D) * res - W / 2
    # exact-tie atoms: on a grid node (distance to neighbours is exactly k*res)
    k = min(6, n_atoms)
    nodes = rng.integers(0, D, (k, 3))
    xyz[:k] = axis[nodes]
    # an atom exactly r=1.0 beyond the box face (box cull is strict, membership inclusive)
    xyz[k] = [W / 2 + 1.0, axis[D // 2], axis[D // 3]]
    xyz[k + 1] = [axis[2], -W / 2 - 1.0, axis[1]]
    # atoms close to reference-block seams (first plane of a block +- r)
    xyz[k + 2] = [axis[min(8, D - 1)] - 1.0 + 0.25 * res, axis[3], axis[4]]
    xyz[k + 3] = [axis[min(8, D - 1)] + 0.5 * res - 1.0, axis[5] + 0.1, axis[min(8, D - 1)] - 1.0 + 0.5 * res]
    feats = rng.random((n_atoms, n_chan)).astype(np.float32)
    feats[rng.random((n_atoms, n_chan)) < 0.3] = 0.0
    types = rng.integers(0, n_chan, n_atoms).astype(np.int16)
    types[0] = n_chan - 1  # make max(types)+1 == n_chan
    r_atom = rng.uniform(0.7, 1.8, n_atoms).astype(np.float32)
    r_atom[:k] = np.float32(1.0)  # ties stay ties
    r_chan = rng.uniform(0.8, 1.6, n_chan).astype(np.float32)
    r_chan[0] = np.float32(1.0)
    return xyz, feats, types, r_atom, r_chan


def gen_small(molvoxel):
    geoms = [  # (D, res, blockdim or None for default 8, n_atoms)
        (16, 0.5, None, 40),
        (16, 0.5, 16, 40),
        (24, 0.4, None, 90),
        (20, 0.5, None, 60),
        (12, 0.75, 5, 30),
        (32, 0.5, None, 400),
    ]
    dens = [("gaussian", 0.5), ("binary", 0.5), ("gaussian", 1.0)]
    store, index = {}, []
    rng = np.random.default_rng(20240)
    for gi, (D, res, bd, n_atoms) in enumerate(geoms):
        C = 5
        xyz, feats, types, r_atom, r_chan = small_inputs(rng, D, res, n_atoms, C)
        store[f"g{gi}/coords"], store[f"g{gi}/features"], store[f"g{gi}/types"] = xyz, feats, types
        store[f"g{gi}/r_atom"], store[f"g{gi}/r_chan"] = r_atom, r_chan
        for density, sigma in dens:
            if gi == 5 and sigma == 1.0:
                continue
            kw = {} if bd is None else {"blockdim": bd}
            for radii_type, rad in (("scalar", 1.0), ("atom-wise", r_atom), ("channel-wise", r_chan)):
                v = molvoxel.create_voxelizer(res, D, radii_type, density, "numpy", sigma=sigma, **kw)
                for mode, chan in (("features", feats), ("types", types), ("single", None)):
                    if mode == "single" and radii_type == "channel-wise":
                        continue
                    if gi == 5 and (radii_type == "channel-wise" or mode == "single"):
                        continue
                    out = v.forward(xyz, None, chan, rad)
                    cid = f"g{gi}_{density}{sigma}_{radii_type}_{mode}"
                    store[f"{cid}/out"] = out
                    index.append(
                        dict(id=cid, geom=gi, dimension=D, resolution=res, blockdim=bd, density=density,
                             sigma=sigma, radii_type=radii_type, mode=mode, scalar_radius=1.0)
                    )
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "small_cases.npz"), **store)
    print(f"small_cases: {len(index)} cases")


def gen_p64(molvoxel):
    """precision=64 (numpy/voxelizer.py:33-34): float64 grids from the reference on small inputs."""
    geoms = [(16, 0.5, None, 40), (12, 0.75, 5, 30), (20, 0.5, None, 60)]
    store, index = {}, []
    rng = np.random.default_rng(6464)
    for gi, (D, res, bd, n_atoms) in enumerate(geoms):
        C = 5
        xyz, feats, types, r_atom, r_chan = small_inputs(rng, D, res, n_atoms, C)
        store[f"g{gi}/coords"], store[f"g{gi}/features"], store[f"g{gi}/types"] = xyz, feats, types
        store[f"g{gi}/r_atom"], store[f"g{gi}/r_chan"] = r_atom, r_chan
        kw = {} if bd is None else {"blockdim": bd}
        for density, sigma in (("gaussian", 0.5), ("binary", 0.5)):
            for radii_type, rad in (("scalar", 1.0), ("atom-wise", r_atom), ("channel-wise", r_chan)):
                v = molvoxel.create_voxelizer(res, D, radii_type, density, "numpy", sigma=sigma, precision=64, **kw)
                for mode, chan in (("features", feats), ("types", types), ("single", None)):
                    if mode == "single" and radii_type == "channel-wise":
                        continue
                    if gi == 2 and not (radii_type == "atom-wise" and mode != "single"):
                        continue
                    out = v.forward(xyz, None, chan, rad)
                    assert out.dtype == np.float64
                    cid = f"p64_g{gi}_{density}_{radii_type}_{mode}"
                    store[f"{cid}/out"] = out
                    index.append(dict(id=cid, geom=gi, dimension=D, resolution=res, blockdim=bd, density=density,
                                      sigma=sigma, radii_type=radii_type, mode=mode, scalar_radius=1.0))
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "p64_cases.npz"), **store)
    print(f"p64_cases: {len(index)} cases")


def gen_dense(molvoxel):
    """Dense clusters: hundreds of in-radius atoms per voxel, so that float32 sums are large and the order of
    summation (reference: BLAS sgemm, unspecified) matters. Full arrays from the reference; they pin the size of the
    HIP-vs-reference error where it is largest (tests/tolerance.py states the rule the tests apply)."""
    store, index = {}, []
    rng = np.random.default_rng(8086)

    def add(cid, D, density, sigma, radii_type, mode, xyz, chan, rad):
        v = molvoxel.create_voxelizer(0.5, D, radii_type, density, "numpy", sigma=sigma)
        out = v.forward(xyz, None, chan, rad)
        store[f"{cid}/coords"], store[f"{cid}/out"] = xyz, out
        if chan is not None:
            store[f"{cid}/chan"] = chan
        if not np.isscalar(rad):
            store[f"{cid}/radii"] = rad
        index.append(dict(id=cid, dimension=D, density=density, sigma=sigma, radii_type=radii_type, mode=mode,
                          scalar_radius=float(rad) if np.isscalar(rad) else None, max=float(out.max())))
        print(cid, out.shape, "max", float(out.max()))

    xyz = rng.normal(scale=0.8, size=(3000, 3))
    add("d0_cluster_features", 24, "gaussian", 0.5, "scalar", "features", xyz, rng.random((3000, 32)).astype(np.float32), 1.0)
    add("d1_cluster_types", 24, "gaussian", 0.5, "scalar", "types", xyz, rng.integers(0, 3, 3000).astype(np.int16), 1.0)
    add("d2_cluster_single_binary", 24, "binary", 0.5, "scalar", "single", xyz, None, 1.0)
    W = 0.5 * 31
    xyz2 = rng.uniform(-W / 2, W / 2, (2000, 3))
    add("d3_uniform_r1.5_features", 32, "gaussian", 0.5, "scalar", "features", xyz2, rng.random((2000, 8)).astype(np.float32), 1.5)
    add("d4_uniform_atomwise_features", 32, "gaussian", 1.0, "atom-wise", "features", xyz2,
        rng.random((2000, 8)).astype(np.float32), rng.uniform(1.0, 2.0, 2000).astype(np.float32))
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "dense_cases.npz"), **store)
    print("dense_cases:", len(index))


# ----------------------------------------------------------------------------- big cases
def run_ref(molvoxel, wl, i=0, density=None, mode=None):
    v = molvoxel.create_voxelizer(
        wl.resolution, wl.dimension, wl.radii_type, density or wl.density, "numpy", sigma=wl.sigma
    )
    mode = mode or wl.mode
    chan = None if mode == "single" else wl.channels[i]
    return v.forward(wl.coords[i], wl.centers[i], chan, wl.radii[i])


def summarise(out, rng_seed):
    flat = out.reshape(-1)
    idx = np.random.default_rng(rng_seed).integers(0, flat.size, 10000)
    nzi = np.flatnonzero(flat)
    if nzi.size:  # half of the samples from non-zero voxels
        idx[:5000] = nzi[np.random.default_rng(rng_seed + 1).integers(0, nzi.size, 5000)]
    return dict(
        sha256=hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest(),
        chan_sums=out.reshape(out.shape[0], -1).sum(axis=1, dtype=np.float64),
        nonzero=int(np.count_nonzero(flat)),
        sample_idx=idx.astype(np.int64),
        sample_val=flat[idx].copy(),
    )


def gen_big(molvoxel, pc):
    from molvoxel_amd import workloads as W

    store, index = {}, []

    def add(cid, wl, i, out, exact):
        s = summarise(out, 77)
        for k in ("chan_sums", "sample_idx", "sample_val"):
            store[f"{cid}/{k}"] = s[k]
        index.append(dict(id=cid, workload=wl.name, molecule=i, sha256=s["sha256"], nonzero=s["nonzero"], exact=exact,
                          shape=list(out.shape)))
        print(cid, out.shape, "nonzero", s["nonzero"])

    w1 = W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"])
    add("cfg1_features_gaussian", w1, 0, run_ref(molvoxel, w1), False)
    w2 = W.cfg2()
    add("cfg2_features_gaussian", w2, 0, run_ref(molvoxel, w2), False)
    add("cfg2_single_binary", w2, 0, run_ref(molvoxel, w2, density="binary", mode="single"), True)
    w3 = W.cfg3()
    add("cfg3_types_binary", w3, 0, run_ref(molvoxel, w3), True)
    add("cfg3_types_gaussian", w3, 0, run_ref(molvoxel, w3, density="gaussian"), False)
    w4 = W.cfg4(batch=8)
    for i in range(8):
        add(f"cfg4_features_gaussian_m{i}", w4, i, run_ref(molvoxel, w4, i), False)
    w5 = W.cfg5()
    add("cfg5_features_gaussian", w5, 0, run_ref(molvoxel, w5), False)
    add("cfg5_single_binary", w5, 0, run_ref(molvoxel, w5, density="binary", mode="single"), True)
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "big_cases.npz"), **store)


# ----------------------------------------------------------------------------- transforms & API
def gen_transforms(molvoxel):
    from molvoxel.voxelizer.numpy.transform import RandomTransform, do_random_transform

    rng = np.random.default_rng(5)
    xyz = rng.normal(size=(20, 3)) * 4.0
    center = xyz.mean(axis=0)
    store, index = {"coords": xyz, "center": center}, []
    for k, (seed, trans, rot, use_center) in enumerate(
        [(1, 0.0, True, True), (2, 0.5, False, True), (3, 0.5, True, True), (4, 1.25, True, False), (5, 0.0, False, False)]
    ):
        np.random.seed(seed)
        out = do_random_transform(xyz, center if use_center else None, trans, rot)
        store[f"t{k}/out"] = out
        after = np.random.rand(2)  # pins how many draws were consumed
        store[f"t{k}/next_rand"] = after
        index.append(dict(id=f"t{k}", seed=seed, random_translation=trans, random_rotation=rot, use_center=use_center))
    # T objects: get_transform() draws translation first, then the quaternion
    for k, (seed, trans, rot) in enumerate([(11, 0.5, True), (12, 0.0, True), (13, 0.7, False)]):
        np.random.seed(seed)
        T = RandomTransform(trans, rot).get_transform()
        store[f"T{k}/out"] = T(xyz, center)
        store[f"T{k}/translation"] = np.zeros((1, 3), np.float32) if T.translation is None else T.translation
        store[f"T{k}/quaternion"] = np.zeros(4) if T.quaternion is None else np.array(T.quaternion)
        index.append(dict(id=f"T{k}", seed=seed, random_translation=trans, random_rotation=rot))
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "transform_cases.npz"), **store)
    print("transform_cases:", len(index))


def gen_api(molvoxel, pc):
    """End-to-end calls through the reference's public API on the 10gs point clouds:
    centring, seeded random transform inside forward(), out_grid reuse, property switches."""
    store, index = {}, []
    lig, lig_t, lig_f = pc["ligand_xyz"], pc["ligand_types"], pc["ligand_feat5"]
    center = lig.mean(axis=0)
    poc, poc_t = pc["pocket_xyz"], pc["pocket_types"]
    sys_xyz = np.concatenate([lig, poc])
    sys_t = np.concatenate([lig_t, poc_t + 5]).astype(np.int16)  # ligand channels 0-4, pocket 5-9

    v = molvoxel.create_voxelizer(0.5, 32, library="numpy")
    g = v.get_empty_grid(10)  # one channel more than max(types)+1: allowed for types
    out = v.forward(sys_xyz, center, sys_t, 1.0, out_grid=g)
    assert out is g
    store["a0/out"] = out.copy()
    index.append(dict(id="a0", what="types scalar gaussian 32^3, ligand+pocket, center=ligand centroid"))

    np.random.seed(123)
    out = v.forward(sys_xyz, center, sys_t, 1.0, random_translation=0.5, random_rotation=True)
    store["a1/out"] = out
    index.append(dict(id="a1", what="same with np.random.seed(123), random_translation=0.5, random_rotation=True"))

    v.radii_type = "channel-wise"
    r_chan = np.array([1.7, 1.55, 1.52, 1.8, 1.6, 1.7, 1.55, 1.52, 1.8, 1.6], dtype=np.float32)[: int(sys_t.max()) + 1]
    store["a2/r_chan"] = r_chan
    store["a2/out"] = v.forward(sys_xyz, center, sys_t, r_chan)
    index.append(dict(id="a2", what="types channel-wise radii"))

    v.radii_type = "atom-wise"
    r_atom = r_chan[sys_t]
    store["a3/out"] = v.forward(sys_xyz, center, sys_t, r_atom)
    index.append(dict(id="a3", what="types atom-wise radii (= channel radii gathered)"))

    v.radii_type = "scalar"
    v.density_type = "binary"
    store["a4/out"] = v.forward(sys_xyz, center, sys_t, 1.5)
    index.append(dict(id="a4", what="types binary scalar r=1.5"))

    v2 = molvoxel.create_voxelizer(0.5, 32, library="numpy")
    store["a5/out"] = v2.forward(lig, center, lig_f, 1.0)
    index.append(dict(id="a5", what="ligand features C=5 gaussian"))
    store["a6/out"] = v2.forward(lig, center, None, 1.0)
    index.append(dict(id="a6", what="ligand single gaussian"))
    store["a7/out"] = v2.forward(lig.astype(np.float32), None, lig_f, 1.0)
    index.append(dict(id="a7", what="float32 coords, center=None (ligand mostly outside the box)"))

    store["sys_xyz"], store["sys_types"], store["center"] = sys_xyz, sys_t, center
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(GOLD, "api_cases.npz"), **store)
    print("api_cases:", len(index))


def gen_dx(ref_root):
    """DX text dumps written by the reference's own writer. molvoxel/etc/pymol/__init__.py needs PyMOL, so dx.py is
    loaded on its own from its file."""
    import importlib.util
    import tempfile

    spec = importlib.util.spec_from_file_location("ref_dx", os.path.join(ref_root, "molvoxel", "etc", "pymol", "dx.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(3)
    cases = {}
    for name, shape, res, cen in [("a", (4, 5, 6), 0.5, (1.0, -2.5, 3.25)), ("b", (3, 3, 3), 0.4, (0.0, 0.0, 0.0)),
                                  ("c", (2, 2, 5), 0.75, (10.0, 20.0, -30.0))]:
        v = (rng.random(shape) * 3 - 1).astype(np.float32)
        with tempfile.NamedTemporaryFile("r", suffix=".dx") as tf:
            mod.write_grid_to_dx_file(tf.name, v, cen, res)
            txt = open(tf.name).read()
        cases[f"{name}/values"], cases[f"{name}/center"] = v, np.array(cen)
        cases[f"{name}/resolution"], cases[f"{name}/text"] = np.array(res), np.array(txt)
    np.savez_compressed(os.path.join(GOLD, "dx_cases.npz"), **cases)
    print("dx_cases:", len(cases) // 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.ref, "molvoxel")):
        sys.exit("reference not present: goldens can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    import molvoxel  # the reference

    os.makedirs(GOLD, exist_ok=True)
    d10 = os.path.join(args.ref, "test", "10gs")
    lig_xyz, lig_sym, lig_arom = read_sdf_heavy(os.path.join(d10, "10gs_ligand.sdf"))
    prot_xyz, prot_sym = read_pdb_heavy(os.path.join(d10, "10gs_protein_nowater.pdb"))
    center = lig_xyz.mean(axis=0)
    near = (np.abs(prot_xyz - center) < 0.5 * 63 / 2 + 1.0).all(axis=1)  # inside the 64^3 box + 1 A
    lig_t = element_types(lig_sym)
    feat5 = np.zeros((lig_xyz.shape[0], 5), dtype=np.float32)
    for c in range(4):
        feat5[:, c] = lig_t == c
    feat5[:, 4] = lig_arom
    pc = dict(
        ligand_xyz=lig_xyz, ligand_types=lig_t, ligand_feat5=feat5, pocket_xyz=prot_xyz[near],
        pocket_types=element_types([s for s, k in zip(prot_sym, near) if k]),
        protein_xyz=prot_xyz, protein_types=element_types(prot_sym),  # the whole chain: what test_time_*.py voxelizes
    )
    np.savez_compressed(os.path.join(GOLD, "pointcloud_10gs.npz"), **pc)
    print("10gs: ligand heavy", lig_xyz.shape[0], "protein heavy", prot_xyz.shape[0], "pocket", int(near.sum()))

    only = set(filter(None, args.only.split(",")))
    if not only or "small" in only:
        gen_small(molvoxel)
    if not only or "p64" in only:
        gen_p64(molvoxel)
    if not only or "dense" in only:
        gen_dense(molvoxel)
    if not only or "big" in only:
        gen_big(molvoxel, pc)
    if not only or "transform" in only:
        gen_transforms(molvoxel)
    if not only or "api" in only:
        gen_api(molvoxel, pc)
    if not only or "dx" in only:
        gen_dx(args.ref)


if __name__ == "__main__":
    main()
