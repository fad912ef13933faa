"""The one tolerance rule of the parity tests.

north_star (BASELINE.json): binary bit-exact; Gaussian within 1e-5 absolute. A Gaussian voxel is a float32 sum of
n in-radius terms, each in (e^-2, 1] times a feature; where hundreds of atoms overlap (dense clusters) the sums reach
10^2..10^3 and one float32 ulp of the result alone is ~1e-5, so an absolute bar is meaningless there. The rule, applied
per voxel:

    |out - ref| <= GAUSS_TOL * max(1, |ref|)          GAUSS_TOL = 5e-6

i.e. absolute 5e-6 (half the north-star bar) for sums up to 1 and relative 5e-6 above. The BASELINE configurations
themselves (cfg-1/2/4/5: sums of a few units) are held to the north-star bar as written - 1e-5 absolute on the full
array - by assert_north_star. Measured against the
reference's own outputs on the dense goldens (tests/golden/dense_cases.npz, sums up to 197): C oracle <= 1.6e-6
relative; the HIP path's figure is recorded in DESIGN.md §4. Membership (which voxels are non-zero) must always be
identical, and binary types/single grids bit-identical.
"""
import numpy as np

GAUSS_TOL = 5e-6
NORTH_STAR_ABS = 1e-5  # BASELINE.json north_star: "within 1e-5 abs for the Gaussian kernel" - asserted as such, full array,
                       # on every BASELINE configuration (sums of a few units); the relative branch above is for the dense
                       # goldens and the fuzz's clustered draws only
P64_TOL = 1e-12  # float64 grids: exp / summation-order differences only


def assert_membership(out, ref):
    bad = int(np.not_equal(out != 0, ref != 0).sum())
    assert bad == 0, f"membership differs in {bad} voxels"


def gaussian_excess(out, ref, tol=GAUSS_TOL):
    """max over voxels of |out - ref| / (tol * max(1, |ref|)); <= 1 passes."""
    ref = np.asarray(ref)
    return float((np.abs(np.asarray(out) - ref) / (tol * np.maximum(1.0, np.abs(ref)))).max()) if ref.size else 0.0


def assert_gaussian(out, ref, tol=GAUSS_TOL):
    assert out.shape == ref.shape
    assert_membership(out, ref)
    ex = gaussian_excess(out, ref, tol)
    assert ex <= 1.0, f"|out-ref| exceeds {tol} * max(1,|ref|) by a factor {ex:.3g}"


def assert_exact(out, ref):
    assert out.shape == ref.shape
    assert np.array_equal(out, ref)


def assert_north_star(out, ref):
    """The north-star bar itself: membership identical and |out - ref| <= 1e-5 ABSOLUTE on every voxel. Returns max |d|."""
    assert out.shape == ref.shape
    assert_membership(out, ref)
    d = np.abs(np.asarray(out, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    worst = float(d.max()) if d.size else 0.0
    if worst > NORTH_STAR_ABS:
        at = np.unravel_index(int(d.argmax()), d.shape)
        raise AssertionError(f"|out-ref| = {worst:.3g} > 1e-5 at voxel {at}: out {out[at]!r}, ref {ref[at]!r}")
    return worst
