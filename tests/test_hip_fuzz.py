"""Randomised parity sweep of the HIP path against the CPU oracle.

240 seeded configurations over everything the operator accepts: grid size (incl. sizes that are not multiples of
the sub-tile edges), resolution, reference blockdim (incl. ones that do not divide the sub-tiles), density, sigma,
radii type, operator, channel count (1 … 40: several channel chunks), atom count (0 … 3000, clustered or spread,
inside and outside the box, on grid nodes), centring and random transforms, host and device inputs.
Bar as everywhere (tests/tolerance.py): membership identical; binary types/single bit-exact; Gaussian and feature sums
within 5e-6 * max(1, |ref|) per voxel. A float64 handle is swept against the numpy port the same way (1e-12).
"""
import numpy as np
import pytest

from tests.tolerance import GAUSS_TOL, P64_TOL, assert_gaussian

pytestmark = pytest.mark.gpu

N_CASES = 240


DIMS = [5, 8, 11, 16, 17, 24, 31, 32, 33, 40, 48, 50, 64, 70]  # (tools/soak.py SOAK_DIMS=... draws from other sizes)


def _draw(seed):
    rng = np.random.default_rng(10_000 + seed)
    D = int(rng.choice(DIMS))
    res = float(rng.choice([0.3, 0.4, 0.5, 0.75, 1.0]))
    blockdim = rng.choice([None, None, 4, 5, 8, 12, 16, D])
    blockdim = None if blockdim is None else int(blockdim)
    density = str(rng.choice(["gaussian", "binary"]))
    sigma = float(rng.choice([0.3, 0.5, 1.0]))
    radii_type = str(rng.choice(["scalar", "atom-wise", "channel-wise"]))
    mode = str(rng.choice(["features", "types", "single"]))
    if mode == "single" and radii_type == "channel-wise":
        radii_type = "atom-wise"
    C_ = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 32, 33, 40]))
    N = int(rng.choice([0, 1, 2, 7, 33, 64, 65, 200, 700, 3000]))
    W = res * (D - 1)
    style = str(rng.choice(["spread", "cluster", "shell"]))
    if style == "spread":
        xyz = rng.uniform(-W / 2 - 2.0, W / 2 + 2.0, (N, 3))
    elif style == "cluster":  # hundreds of candidates per slab: extension lines / x-list path
        xyz = rng.normal(0.0, max(0.3, W / 12), (N, 3)) + rng.uniform(-W / 4, W / 4, 3)
    else:  # atoms around the faces of the box: cull edge cases
        xyz = rng.uniform(-W / 2, W / 2, (N, 3))
        ax = rng.integers(0, 3, N)
        xyz[np.arange(N), ax] = rng.choice([-1.0, 1.0], N) * (W / 2 + rng.uniform(-1.2, 1.2, N))
    k = min(N, 5)  # exact ties: atoms on grid nodes
    if k:
        xyz[:k] = (rng.integers(0, D, (k, 3)) * res - W / 2)
    center = rng.uniform(-3, 3, 3) if rng.random() < 0.5 else None
    if center is not None:
        xyz = xyz + center
    feats = rng.random((N, C_)).astype(np.float32)
    feats[rng.random((N, C_)) < 0.3] = 0.0
    types = rng.integers(0, C_, N).astype(np.int16)
    if N:
        types[0] = C_ - 1
    base_r = float(rng.choice([0.6, 1.0, 1.5, 2.2])) * max(res / 0.5, 0.6)
    r_atom = (base_r * rng.uniform(0.6, 1.4, N)).astype(np.float32)
    r_atom[:k] = np.float32(2 * res)  # ties stay ties
    r_chan = (base_r * rng.uniform(0.6, 1.4, C_)).astype(np.float32)
    radii = {"scalar": base_r, "atom-wise": r_atom, "channel-wise": r_chan}[radii_type]
    chan = {"features": feats, "types": types, "single": None}[mode]
    return dict(D=D, res=res, blockdim=blockdim, density=density, sigma=sigma, radii_type=radii_type, mode=mode, C=C_,
                N=N, xyz=xyz, center=center, chan=chan, radii=radii, device=bool(rng.random() < 0.5), seed=seed)


def _reference(case, coords_for_oracle, precision):
    from oracle import c_oracle, numpy_port

    nch = {"features": case["C"], "types": case["C"], "single": 1}[case["mode"]]
    if case["N"] == 0:
        return np.zeros((nch, case["D"], case["D"], case["D"]), np.float32 if precision == 32 else np.float64)
    kw = dict(radii_type=case["radii_type"], density=case["density"], sigma=case["sigma"], num_channels=nch)
    if precision == 32:
        return c_oracle.voxelize(coords_for_oracle, case["chan"], case["radii"], resolution=case["res"],
                                 dimension=case["D"], blockdim=case["blockdim"], **kw)
    spec = numpy_port.GridSpec(case["res"], case["D"], case["blockdim"])
    return numpy_port.voxelize(spec, coords_for_oracle, case["chan"], case["radii"], precision=64, **kw)


def _run(mv, case, precision):
    import torch

    extra = {} if case["blockdim"] is None else {"blockdim": case["blockdim"]}
    v = mv.create_voxelizer(case["res"], case["D"], case["radii_type"], case["density"], "hip", sigma=case["sigma"],
                            precision=precision, output="torch" if case["device"] else "numpy", **extra)
    coords, center, chan, radii = case["xyz"], case["center"], case["chan"], case["radii"]
    if case["device"]:
        coords_in, chan_in = v.asarray(coords, "coords"), (None if chan is None else v.asarray(chan, case["mode"]))
        radii_in = radii if np.isscalar(radii) else v.asarray(radii, "radii")
        center_in = None if center is None else v.asarray(center, "center")
    else:
        coords_in, chan_in, radii_in, center_in = coords, chan, radii, center
    nch = {"features": case["C"], "types": case["C"], "single": 1}[case["mode"]]
    grid = v.get_empty_grid(nch)
    if case["device"]:
        grid.fill_(7.0)  # stale content must be overwritten
    else:
        grid.fill(7.0)
    # both routes through the library: the binned three-launch pipeline and the single-launch direct kernel must
    # produce the same bits (same candidates, same atom order)
    v.debug_option("direct", 0)
    out = v.forward(coords_in, center_in, chan_in, radii_in, out_grid=grid)
    assert out is grid
    out = out.cpu().numpy().copy() if isinstance(out, torch.Tensor) else out.copy()
    if precision == 32:
        v.debug_option("direct", 1)
        if case["device"]:
            grid.fill_(3.0)
        else:
            grid.fill(3.0)
        again = v.forward(coords_in, center_in, chan_in, radii_in, out_grid=grid)
        again = again.cpu().numpy() if isinstance(again, torch.Tensor) else again
        assert np.array_equal(out, again), "direct kernel and binned pipeline disagree"
    return out, (coords - center if center is not None else coords)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_configuration(seed):
    import molvoxel_amd as mv

    case = _draw(seed)
    if case["N"] == 0 and case["mode"] == "types":
        pytest.skip("max(types) of an empty array: the reference raises as well (numpy/voxelizer.py:278)")
    precision = 64 if seed % 8 == 7 else 32
    out, moved = _run(mv, case, precision)
    ref = _reference(case, moved, precision)
    assert out.shape == ref.shape and out.dtype == ref.dtype
    bad = np.not_equal(out != 0, ref != 0).sum()
    assert bad == 0, f"membership differs in {bad} voxels: { {k: v for k, v in case.items() if k not in ('xyz', 'chan', 'radii')} }"
    if case["density"] == "binary" and case["mode"] != "features":
        assert np.array_equal(out, ref)
    else:
        assert_gaussian(out, ref, GAUSS_TOL if precision == 32 else P64_TOL)


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_with_random_transform(seed):
    """Same sweep with the reference's in-call random transform: the host draws (quaternion, translation) from the
    global numpy RNG in the reference's order; the oracle gets the coordinates `do_random_transform` produces from
    the same RNG state (numpy/transform.py:63-80 restated in molvoxel_amd/voxelizer/hip/transform.py)."""
    import molvoxel_amd as mv
    from molvoxel_amd.voxelizer.hip.transform import do_transform, draw_forward_transform

    case = _draw(500 + seed)
    if case["N"] == 0:
        pytest.skip("empty molecule")
    extra = {} if case["blockdim"] is None else {"blockdim": case["blockdim"]}
    v = mv.create_voxelizer(case["res"], case["D"], case["radii_type"], case["density"], "hip", sigma=case["sigma"],
                            output="numpy", **extra)
    nch = {"features": case["C"], "types": case["C"], "single": 1}[case["mode"]]
    v.debug_option("direct", seed % 2)  # the transform runs in prep_kernel (binned) or in the scan (direct)
    np.random.seed(seed)
    out = v.forward(case["xyz"], case["center"], case["chan"], case["radii"], 0.7, True, out_grid=v.get_empty_grid(nch))
    # replay: same RNG state -> same draw -> the reference's transform arithmetic on the host (float64)
    np.random.seed(seed)
    translation, quaternion = draw_forward_transform(0.7, True)
    moved = case["xyz"] - case["center"] if case["center"] is not None else case["xyz"]
    moved = do_transform(moved, None, translation, quaternion)
    ref = _reference(case, moved, 32)
    assert np.array_equal(out != 0, ref != 0)
    if case["density"] == "binary" and case["mode"] != "features":
        assert np.array_equal(out, ref)
    else:
        assert_gaussian(out, ref)


@pytest.mark.parametrize("seed", range(40))
def test_random_batches(seed):
    """forward_batch over ragged molecules (empty ones included) with per-molecule centres, every operator; each
    molecule's grid must equal the oracle's for that molecule alone. Big grids and long atom lists are in the draw."""
    import molvoxel_amd as mv
    from oracle import c_oracle

    rng = np.random.default_rng(77_000 + seed)
    D = int(rng.choice([16, 24, 33, 48, 64, 96, 128]))
    res = float(rng.choice([0.4, 0.5, 1.0]))
    B = int(rng.choice([1, 2, 5, 9, 20])) if D <= 64 else int(rng.choice([1, 2, 3]))
    mode = str(rng.choice(["features", "types", "single"]))
    radii_type = str(rng.choice(["scalar", "atom-wise"] + ([] if mode == "single" else ["channel-wise"])))
    density = str(rng.choice(["gaussian", "binary"]))
    C_ = 1 if mode == "single" else int(rng.choice([1, 4, 6, 16, 32, 35]))
    if D > 64:
        C_ = min(C_, 6)
    W = res * (D - 1)
    sizes = [int(rng.choice([0, 1, 40, 300, 2000, 12000])) for _ in range(B)]
    coords = [rng.uniform(-W / 2 - 1, W / 2 + 1, (n, 3)) for n in sizes]
    centers = rng.uniform(-2, 2, (B, 3))
    coords = [c + centers[b] for b, c in enumerate(coords)]
    feats = [rng.random((n, C_)).astype(np.float32) for n in sizes]
    types = [rng.integers(0, C_, n).astype(np.int16) for n in sizes]
    r_atom = [rng.uniform(0.8, 2.0, n).astype(np.float32) * res / 0.5 for n in sizes]
    r_chan = (rng.uniform(0.8, 2.0, C_) * res / 0.5).astype(np.float32)
    chan = None if mode == "single" else np.concatenate(feats if mode == "features" else types)
    radii = {"scalar": 1.3 * res / 0.5, "atom-wise": np.concatenate(r_atom), "channel-wise": r_chan}[radii_type]
    offsets = np.cumsum([0] + sizes)
    v = mv.create_voxelizer(res, D, radii_type, density, "hip", sigma=0.6, output="numpy")
    v.debug_option("direct", 0)
    out = v.forward_batch(np.concatenate(coords), offsets, centers, chan, radii, num_channels=C_).copy()
    assert out.shape == (B, C_, D, D, D)
    v.debug_option("direct", 1)
    assert np.array_equal(out, v.forward_batch(np.concatenate(coords), offsets, centers, chan, radii, num_channels=C_)), \
        "direct kernel and binned pipeline disagree"
    for b, n in enumerate(sizes):
        if n == 0:
            assert not out[b].any()
            continue
        ch = None if mode == "single" else (feats[b] if mode == "features" else types[b])
        rad = {"scalar": radii, "atom-wise": r_atom[b], "channel-wise": r_chan}[radii_type]
        ref = c_oracle.voxelize(coords[b] - centers[b], ch, rad, resolution=res, dimension=D, radii_type=radii_type,
                                density=density, sigma=0.6, num_channels=C_)
        assert np.array_equal(out[b] != 0, ref != 0), (seed, b)
        if density == "binary" and mode != "features":
            assert np.array_equal(out[b], ref), (seed, b)
        else:
            assert_gaussian(out[b], ref)
