"""The per-molecule launch (voxelize_pair_kernel, molvoxel_amd/csrc/mvx_pair.hip) at the edges of its own bookkeeping.

The kernel replaces the body of one `forward()` call of the reference (numpy/voxelizer.py:97-169, 240-315, 370-436) with a
float32 scan -> LDS stash -> float64 stage -> walk pipeline; every case below forces that route ("direct" test option), checks
it against the CPU oracle on the same inputs (membership identical, Gaussian within tests/tolerance.py) and asserts that the
binned pipeline gives the same BITS (both routes add the same float32 terms in atom order).
Covered: survivors clustered in one wave's share (beyond the 48-entry stash: the trip back to memory; beyond 64 and 256 rows:
several staging waves / rounds), odd atom counts at every block boundary (the molecule's last 16-byte chunk), molecules of
several segments, the smallest grids (one or two waves per slab), per-type radii beyond the LDS table, ragged batches,
transforms with a device-resident centre, degenerate radii.
"""
import numpy as np
import pytest

from tests.tolerance import assert_exact, assert_gaussian

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mv():
    import molvoxel_amd

    return molvoxel_amd


def _both_routes(v, call):
    outs = []
    for route in (1, 0):  # one launch, then binned
        v.debug_option("direct", route)
        outs.append(np.array(call(), copy=True))
    v.debug_option("direct", -1)
    assert np.array_equal(outs[0], outs[1]), "one-launch and binned routes differ in bits"
    return outs[0]


def _check(out, ref, density):
    if density == "binary":
        assert np.array_equal(out != 0, ref != 0)
        assert np.abs(out - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    else:
        assert_gaussian(out, ref)


@pytest.mark.parametrize("cluster", [40, 60, 100, 300, 600])
@pytest.mark.parametrize("density", ["gaussian", "binary"])
def test_survivors_clustered_in_one_waves_share(mv, cluster, density):
    """`cluster` consecutive atoms inside a 1.2 A ball (residue-like locality): one scanning wave finds them all - more than
    its stash holds (48), more than one staging wave takes (64), more than a round of rows (256, together with the other waves' finds)."""
    from oracle import c_oracle

    rng = np.random.default_rng(100 + cluster)
    D, C_, n = 32, 8, 3000
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2, W_ / 2, (n, 3))
    k0 = 1111
    xyz[k0:k0 + cluster] = np.array([1.3, -2.1, 0.7]) + rng.normal(0, 0.4, (cluster, 3))
    feats = rng.random((n, C_)).astype(np.float32)
    radii = rng.uniform(0.8, 1.6, n).astype(np.float32)
    v = mv.create_voxelizer(0.5, D, "atom-wise", density, "hip", output="numpy", sigma=0.7)
    out = _both_routes(v, lambda: v.forward_features(xyz, None, feats, radii))
    _check(out, c_oracle.voxelize(xyz, feats, radii, dimension=D, radii_type="atom-wise", density=density, sigma=0.7), density)


@pytest.mark.parametrize("n", [129, 255, 257, 383, 385, 511, 1023, 1025, 2047, 2049, 4095, 4097])
def test_odd_atom_counts_at_block_boundaries(mv, n):
    """The scan fetches 16 bytes per lane; an odd atom count ends half way through such a chunk, which is then fetched 8 bytes
    early and its halves swapped - at every position of the last atom inside a 128-atom block, with the LAST atom the one
    that matters (placed at the grid's centre, everything else far from it)."""
    from oracle import c_oracle

    rng = np.random.default_rng(n)
    D = 16
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2, W_ / 2, (n, 3))
    xyz[-1] = [0.1, -0.2, 0.3]
    xyz[-2] = [W_ / 2 - 0.1, W_ / 2 - 0.2, -W_ / 2 + 0.3]
    types = rng.integers(0, 4, n)
    v = mv.create_voxelizer(0.5, D, "scalar", "binary", "hip", output="numpy")
    out = _both_routes(v, lambda: v.forward_types(xyz, None, types, 1.0))
    assert_exact(out, c_oracle.voxelize(xyz, types, 1.0, dimension=D, density="binary", num_channels=4))


@pytest.mark.parametrize("D,n", [(32, 40000), (64, 70000), (16, 9000)])
def test_molecules_of_several_segments(mv, D, n):
    """More atoms than one segment holds (2 048 per wave: 16 384 at D = 32, 32 768 at D = 64, 8 192 at D = 16): the accumulators
    are carried over scan -> stage -> walk rounds. Sparse enough (a 4x wider cloud than the grid) for the oracle to be quick."""
    from oracle import c_oracle

    rng = np.random.default_rng(D + n)
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-2 * W_, 2 * W_, (n, 3))
    feats = rng.random((n, 4)).astype(np.float32)
    v = mv.create_voxelizer(0.5, D, "scalar", "gaussian", "hip", output="numpy", sigma=0.5)
    out = _both_routes(v, lambda: v.forward_features(xyz, None, feats, 1.0))
    _check(out, c_oracle.voxelize(xyz, feats, 1.0, dimension=D, density="gaussian", sigma=0.5), "gaussian")


@pytest.mark.parametrize("D", [4, 8, 12, 20])
@pytest.mark.parametrize("n", [1, 129, 700])
def test_smallest_grids(mv, D, n):
    """One, two or three 8-voxel sub-tiles per row: workgroups of 128 ... 384 threads, the pair still two slabs wide."""
    from oracle import c_oracle

    rng = np.random.default_rng(10 * D + n)
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2 - 1, W_ / 2 + 1, (n, 3))
    feats = rng.random((n, 5)).astype(np.float32)
    for density in ("gaussian", "binary"):
        v = mv.create_voxelizer(0.5, D, "scalar", density, "hip", output="numpy", sigma=0.5)
        out = _both_routes(v, lambda: v.forward_features(xyz, None, feats, 1.1))
        _check(out, c_oracle.voxelize(xyz, feats, 1.1, dimension=D, density=density, sigma=0.5), density)


@pytest.mark.parametrize("C_", [7, 300])
def test_types_with_per_type_radii_inside_and_beyond_the_lds_table(mv, C_):
    """forward_types with channel-wise radii (the reference's commonest call, etc/rdkit/wrapper.py:38-45): the radius of an atom
    is radii[type] (numpy/voxelizer.py:284-285); the kernel keeps the first 256 table entries in LDS. Out-of-range types never
    contribute; a zero and an infinite radius neither."""
    from oracle import c_oracle

    rng = np.random.default_rng(C_)
    D, n = 24, 900
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2, W_ / 2, (n, 3))
    types = rng.integers(0, C_, n)
    radii = rng.uniform(0.7, 1.7, C_).astype(np.float32)
    radii[3] = 0.0
    radii[5] = np.inf
    for density in ("gaussian", "binary"):
        v = mv.create_voxelizer(0.5, D, "channel-wise", density, "hip", output="numpy", sigma=0.6)
        out = _both_routes(v, lambda: v.forward_types(xyz, None, types, radii))
        ref = c_oracle.voxelize(xyz, types, radii, dimension=D, radii_type="channel-wise", density=density, sigma=0.6, num_channels=C_)
        _check(out, ref, density)


def test_ragged_batch_through_the_one_launch_route(mv):
    """Molecules of very different sizes in one call (offsets on the device, one transform slot per molecule): empty, one atom,
    a ligand (no scan), exactly the no-scan limit and one more, a pocket."""
    from oracle import c_oracle

    rng = np.random.default_rng(77)
    D, C_ = 16, 6
    W_ = 0.5 * (D - 1)
    sizes = [0, 1, 33, 256, 257, 2500, 0, 64]  # (256 rows per round at D = 16: the no-scan limit)
    coords = [rng.uniform(-W_ / 2 - 0.5, W_ / 2 + 0.5, (n, 3)) for n in sizes]
    feats = [rng.random((n, C_)).astype(np.float32) for n in sizes]
    centers = rng.uniform(-0.4, 0.4, (len(sizes), 3))
    offsets = np.cumsum([0] + sizes)
    v = mv.create_voxelizer(0.5, D, "scalar", "gaussian", "hip", output="numpy", sigma=0.5)
    out = _both_routes(v, lambda: v.forward_batch(np.concatenate(coords), offsets, centers, np.concatenate(feats), 0.9))
    for b, n in enumerate(sizes):
        if n == 0:
            assert not out[b].any()
            continue
        _check(out[b], c_oracle.voxelize(coords[b] - centers[b], feats[b], 0.9, dimension=D, density="gaussian", sigma=0.5), "gaussian")


@pytest.mark.parametrize("n", [100, 1500])
def test_random_transform_with_a_device_resident_centre(mv, n):
    """The reference's timing loop (test/test_time_numpy.py:11-15): centre given as a device tensor (never read by the host),
    rotation and translation drawn per call; the scan works on a float32 estimate of the transform, the records on the
    reference's float64 expression tree."""
    import torch

    from molvoxel_amd.voxelizer.hip.transform import do_transform, draw_forward_transform
    from oracle import c_oracle

    rng = np.random.default_rng(n)
    D, C_ = 32, 10
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2, W_ / 2, (n, 3)) + 40.0
    center = np.full(3, 40.0) + rng.uniform(-0.5, 0.5, 3)
    feats = rng.random((n, C_)).astype(np.float32)
    v = mv.create_voxelizer(0.5, D, "scalar", "gaussian", "hip", sigma=0.5)
    d_xyz, d_cen, d_f = v.asarray(xyz, "coords"), v.asarray(center, "center"), v.asarray(feats, "features")
    for seed in (1, 2, 3):
        outs = []
        for route in (1, 0):
            v.debug_option("direct", route)
            np.random.seed(seed)
            outs.append(v.forward(d_xyz, d_cen, d_f, 1.0, 0.5, True).clone())
        v.debug_option("direct", -1)
        assert torch.equal(outs[0], outs[1])
        np.random.seed(seed)
        translation, quaternion = draw_forward_transform(0.5, True)
        moved = do_transform(xyz - center, None, translation, quaternion)
        assert_gaussian(outs[0].cpu().numpy(), c_oracle.voxelize(moved, feats, 1.0, dimension=D, density="gaussian", sigma=0.5))


def test_degenerate_atom_wise_radii(mv):
    """Zero, negative, NaN, infinite and huge radii per atom: `d / r <= 1` of the reference (numpy/voxelizer.py:548-555) is false
    for all but the huge one, which covers the whole grid."""
    import torch

    rng = np.random.default_rng(5)
    D, n = 16, 400
    W_ = 0.5 * (D - 1)
    xyz = rng.uniform(-W_ / 2, W_ / 2, (n, 3))
    feats = rng.random((n, 3)).astype(np.float32)
    radii = rng.uniform(0.8, 1.4, n).astype(np.float32)
    bad = {7: 0.0, 8: -1.0, 9: np.nan, 10: np.inf}
    for i, r in bad.items():
        radii[i] = r
    radii[11] = 100.0
    v = mv.create_voxelizer(0.5, D, "atom-wise", "binary", "hip")
    outs = []
    for route in (1, 0):
        v.debug_option("direct", route)
        outs.append(v.forward_features(v.asarray(xyz, "coords"), None, v.asarray(feats, "features"), v.asarray(radii, "radii")).clone())
    v.debug_option("direct", -1)
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()
    keep = np.ones(n, bool)
    keep[list(bad)] = False
    clean = v.forward_features(v.asarray(xyz[keep], "coords"), None, v.asarray(feats[keep], "features"), v.asarray(radii[keep], "radii"))
    assert torch.equal(outs[0], clean)
    # the huge radius reaches every voxel: channel sums are at least that atom's features everywhere
    assert (outs[0].cpu().numpy().reshape(3, -1).min(axis=1) >= feats[11] - 1e-6).all()


def test_batch_with_a_random_transform_per_molecule(mv):
    """forward_batch draws one transform per molecule, in molecule order (the loop of test/test_time_numpy.py:11-15 as one
    call): both routes give the same bits, and the same bits as per-molecule forward() calls that consume the RNG alike."""
    import torch

    rng = np.random.default_rng(4242)
    D, C_ = 24, 12
    W_ = 0.5 * (D - 1)
    sizes = [200, 0, 33, 1500, 129]
    coords = [rng.uniform(-W_ / 2, W_ / 2, (n, 3)) + 7.0 for n in sizes]
    feats = [rng.random((n, C_)).astype(np.float32) for n in sizes]
    centers = np.full((len(sizes), 3), 7.0) + rng.uniform(-0.3, 0.3, (len(sizes), 3))
    offsets = np.cumsum([0] + sizes)
    v = mv.create_voxelizer(0.5, D, "scalar", "gaussian", "hip", sigma=0.5)
    d_xyz, d_f = v.asarray(np.concatenate(coords), "coords"), v.asarray(np.concatenate(feats), "features")
    d_cen = v.asarray(centers, "center")
    outs = []
    for route in (1, 0):
        v.debug_option("direct", route)
        np.random.seed(99)
        outs.append(v.forward_batch(d_xyz, offsets, d_cen, d_f, 1.1, random_translation=0.5, random_rotation=True).clone())
    v.debug_option("direct", -1)
    assert torch.equal(outs[0], outs[1])
    np.random.seed(99)
    for b, n in enumerate(sizes):
        one = v.forward(v.asarray(coords[b], "coords"), v.asarray(centers[b], "center"), v.asarray(feats[b], "features"), 1.1, 0.5, True)
        assert torch.equal(one, outs[0][b]), b
