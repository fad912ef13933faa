"""The CPU oracle (C restatement and numpy port) against the reference's own outputs.

Goldens were produced by importing the reference numpy backend (oracle/gen_golden.py).
Bar: binary bit-exact; Gaussian <= 1e-6 abs (float32 exp / sgemm summation order only).
"""
import hashlib

import numpy as np
import pytest

from oracle import c_oracle, numpy_port
from tests import goldens

Z_SMALL, IDX_SMALL = goldens.load("small_cases.npz")
GAUSS_TOL = 1e-6


def _kw(case):
    return dict(radii_type=case["radii_type"], density=case["density"], sigma=case["sigma"])


@pytest.mark.parametrize("case", IDX_SMALL, ids=[c["id"] for c in IDX_SMALL])
def test_c_oracle_small(case):
    coords, chan, radii = goldens.small_case_inputs(Z_SMALL, case)
    ref = Z_SMALL[f"{case['id']}/out"]
    out = c_oracle.voxelize(coords, chan, radii, resolution=case["resolution"], dimension=case["dimension"],
                            blockdim=case["blockdim"], num_channels=ref.shape[0], **_kw(case))
    if case["density"] == "binary" and case["mode"] != "features":
        assert np.array_equal(out, ref)
    else:
        assert np.array_equal(out != 0, ref != 0), "membership differs"
        assert np.abs(out - ref).max() <= GAUSS_TOL


@pytest.mark.parametrize("case", IDX_SMALL, ids=[c["id"] for c in IDX_SMALL])
def test_numpy_port_small(case):
    coords, chan, radii = goldens.small_case_inputs(Z_SMALL, case)
    ref = Z_SMALL[f"{case['id']}/out"]
    spec = numpy_port.GridSpec(case["resolution"], case["dimension"], case["blockdim"])
    out = numpy_port.voxelize(spec, coords, chan, radii, num_channels=ref.shape[0], **_kw(case))
    # same library calls in the same order as the reference: identical bits
    assert np.array_equal(out, ref)


Z_BIG, IDX_BIG = goldens.load("big_cases.npz")


def _workload(case):
    from molvoxel_amd import workloads as W

    name = case["workload"]
    if name == "cfg1":
        pc = np.load(goldens.GOLD + "/pointcloud_10gs.npz")
        return W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"])
    if name == "cfg4":
        return W.cfg4(batch=8)
    return getattr(W, name)()


@pytest.mark.parametrize("case", IDX_BIG, ids=[c["id"] for c in IDX_BIG])
def test_c_oracle_big(case):
    wl = _workload(case)
    i = case["molecule"]
    parts = case["id"].split("_")
    mode, density = parts[1], parts[2]
    chan = None if mode == "single" else wl.channels[i]
    xyz = wl.coords[i] - wl.centers[i].reshape(1, 3)
    out = c_oracle.voxelize(xyz, chan, wl.radii[i], resolution=wl.resolution, dimension=wl.dimension,
                            radii_type=wl.radii_type, density=density, sigma=wl.sigma,
                            num_channels=case["shape"][0])
    assert list(out.shape) == case["shape"]
    assert int(np.count_nonzero(out)) == case["nonzero"]
    if case["exact"]:
        assert hashlib.sha256(out.tobytes()).hexdigest() == case["sha256"]
    flat = out.reshape(-1)
    idx, val = Z_BIG[f"{case['id']}/sample_idx"], Z_BIG[f"{case['id']}/sample_val"]
    assert np.abs(flat[idx] - val).max() <= (0 if case["exact"] else 2e-6)
    sums = out.reshape(out.shape[0], -1).sum(axis=1, dtype=np.float64)
    assert np.allclose(sums, Z_BIG[f"{case['id']}/chan_sums"], rtol=1e-6, atol=1e-3)


Z_P64, IDX_P64 = goldens.load("p64_cases.npz")


@pytest.mark.parametrize("case", IDX_P64, ids=[c["id"] for c in IDX_P64])
def test_numpy_port_precision64(case):
    """precision=64 (numpy/voxelizer.py:33-34): float64 distances, densities and sums; same calls, same bits."""
    coords, chan, radii = goldens.small_case_inputs(Z_P64, case)
    ref = Z_P64[f"{case['id']}/out"]
    spec = numpy_port.GridSpec(case["resolution"], case["dimension"], case["blockdim"])
    out = numpy_port.voxelize(spec, coords, chan, radii, num_channels=ref.shape[0], precision=64, **_kw(case))
    assert out.dtype == np.float64 and np.array_equal(out != 0, ref != 0)
    if case["mode"] == "features":  # BLAS walks a strided operand in the reference and a copy here: last-bit sums
        assert np.abs(out - ref).max() <= 1e-15 * max(1.0, float(np.abs(ref).max()))
    else:
        assert np.array_equal(out, ref)
