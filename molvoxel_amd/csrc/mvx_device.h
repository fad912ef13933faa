// mvx_device.h - device-side building blocks shared by the kernel translation units (mvx_prep.hip, mvx_slab.hip,
// mvx_pair.hip, mvx_f64.hip): exact arithmetic of the membership rule, the rigid transform, per-atom preparation
// (culls -> admitted voxel ranges), the one-voxel-per-lane accumulator set (OpsF32) with its write-out paths, and the
// staging / filter / walk steps of a slab round (used through mvx_slab_body.inc).
//
// Replaces, on the device, the whole per-call body of the reference's
//   Voxelizer.forward_features / forward_types / forward_single
//   (molvoxel/voxelizer/numpy/voxelizer.py:97-169, 240-315, 370-436 and the helpers they call:
//    _get_overlap :481-494, _get_overlap_blocks :496-527, _calc_grid :531-560,
//    _set_grid_* :194-236, 344-366, 457-477; transform numpy/transform.py:44-60).
// Not a translation of that code (Python loop over 8^3 blocks, cdist -> (V,512) -> matmul), nor of the torch path.
//
// Exactness: membership float32(float32(sqrt_f64(d2))/r32) <= 1 is equivalent to d2 <= T with
//   y  = largest fp64 whose float32 rounding is <= r32,  T = round_down(y * nextup(y))
// (derivation in DESIGN.md, "Exactness"; checked against 20k radii on the CPU and by tests/test_hip_parity.py).
// d2 is formed exactly like scipy cdist: (dx*dx + dy*dy) + dz*dz in fp64 WITHOUT fma, so every TU that includes this
// header is compiled with -ffp-contract=off and without fast-math (and carries the pragma below).
#pragma once
#include "mvx_internal.h"
#include "mvx_tuning.h"

#include <math.h>
#include <type_traits>
#include <hip/hip_ext.h>

// cdist-order arithmetic must not be fused, whatever flags the TU is built with.
#pragma clang fp contract(off)

namespace mvx {

typedef float float2v __attribute__((ext_vector_type(2)));

#ifdef MVX_DIAG
// Diagnostic builds (tools/ab_build.sh diag "-DMVX_DIAG"): s_memtime stamps of the batched voxelize_kernel, 16 x 8 B per
// workgroup, into a buffer the host hands over with set_diag_buffer() (mvx_slab.hip: only that TU's copy of the pointer is
// ever set); the shipped library has none of this.
static __device__ unsigned long long *g_diag = nullptr;
#define VK_STAMP(i) do { if (g_diag && lane == 0 && (wave == 0 || (i) >= 8)) g_diag[16 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VK_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// exact arithmetic of the membership rule
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ double next_up(double x) { // x > 0 finite
    long long b;
    __builtin_memcpy(&b, &x, 8);
    b += 1;
    __builtin_memcpy(&x, &b, 8);
    return x;
}
__host__ __device__ __forceinline__ double next_down(double x) { // x > 0 finite
    long long b;
    __builtin_memcpy(&b, &x, 8);
    b -= 1;
    __builtin_memcpy(&x, &b, 8);
    return x;
}

// Largest fp64 d2 with float32(float32(sqrt(d2)) / r32) <= 1 (sqrt and division correctly rounded).
// (IEEE operations only - conversions, one product, one fma - so the host evaluates it to the same bits.)
__host__ __device__ inline double d2_threshold(float r32) {
    if (!(r32 > 0.0f) || !(r32 < 3.0e38f)) return -1.0;
    unsigned rb;
    __builtin_memcpy(&rb, &r32, 4);
    const unsigned ub_ = rb + 1u;
    float up;
    __builtin_memcpy(&up, &ub_, 4);
    const double m = 0.5 * ((double)r32 + (double)up); // midpoint between r32 and the next float (exact)
    const bool even = (rb & 1u) == 0u;
    const double y = even ? m : next_down(m); // largest fp64 that rounds (ties-to-even) to <= r32
    const double yp = next_up(y);
    const double hi = y * yp;
    const double lo = fma(y, yp, -hi); // exact residual of the product
    return (lo >= 0.0) ? hi : next_down(hi);
}

// float64 grids: largest fp64 d2 with sqrt_f64(d2) / r <= 1 evaluated in float64 as the reference does
// (numpy/voxelizer.py:548-555 with fp = float64). fl(s / r) <= 1 <=> s <= r for doubles s, r > 0 (s > r puts the quotient
// at least one ulp(r)/r > 2^-53 above 1, which rounds above 1), and fl(sqrt(d2)) <= r <=> d2 < (r + ulp(r)/2)^2 =
// r * nextup(r) + ulp^2/4: the same product-and-residual test as above with y = r.
__host__ __device__ __forceinline__ double d2_threshold64(double r) {
    if (!(r > 0.0) || !(r < 1.0e300)) return -1.0;
    const double rp = next_up(r);
    const double hi = r * rp;
    const double lo = fma(r, rp, -hi);
    return (lo >= 0.0) ? hi : next_down(hi);
}
// float64 gaussian: exp(-0.5 * ((d / r) / sigma)^2) = exp(c * d2), c = -0.5 / (r sigma)^2. One rounding chain instead of
// the reference's sqrt, two divisions and a square: both are within ~4 ulp of the exact argument (|arg| <= 0.5/sigma^2),
// i.e. the values agree to ~1e-15 relative.
__host__ __device__ __forceinline__ double gauss_coeff64(double r, double sigma) {
    const double rs = r * sigma;
    return -0.5 / (rs * rs);
}

__host__ __device__ __forceinline__ float gauss_coeff(float r32, float sigma32) {
    const double rs = (double)r32 * (double)sigma32;
    return (float)(-0.5 * 1.4426950408889634 / (rs * rs));
}

// do_transform in the reference's operation order (numpy/transform.py:44-60, _quaternion.py:24-50).
__device__ inline void apply_xform(const mvx_xform &xf, double &x, double &y, double &z) {
    double c0 = xf.center[0], c1 = xf.center[1], c2 = xf.center[2];
    if (xf.flags & MVX_XF_CENTER_PTR) { // a device-resident centre (the host never saw its value)
        c0 = xf.center_ptr[0];
        c1 = xf.center_ptr[1];
        c2 = xf.center_ptr[2];
    }
    if (xf.flags & MVX_XF_CENTER) {
        x = x - c0;
        y = y - c1;
        z = z - c2;
    }
    const double t0 = (double)xf.trans[0], t1 = (double)xf.trans[1], t2 = (double)xf.trans[2];
    if (xf.flags & MVX_XF_ROTATE) {
        const double q0 = xf.quat[0], q1 = xf.quat[1], q2 = xf.quat[2], q3 = xf.quat[3];
        const double zero = 0.0;
        // qp = q * (0, x, y, z)
        const double a0 = ((q0 * zero - q1 * x) - q2 * y) - q3 * z;
        const double a1 = ((q0 * x + q1 * zero) + q2 * z) - q3 * y;
        const double a2 = ((q0 * y - q1 * z) + q2 * zero) + q3 * x;
        const double a3 = ((q0 * z + q1 * y) - q2 * x) + q3 * zero;
        // qp * q^-1, q^-1 = (q0, -q1, -q2, -q3)
        const double i0 = q0, i1 = q1 * -1, i2 = q2 * -1, i3 = q3 * -1;
        x = ((a0 * i1 + a1 * i0) + a2 * i3) - a3 * i2;
        y = ((a0 * i2 - a1 * i3) + a2 * i0) + a3 * i1;
        z = ((a0 * i3 + a1 * i2) - a2 * i1) + a3 * i0;
        if (xf.flags & MVX_XF_RECENTER) { // `coords += center` (numpy/transform.py:53)
            x += c0;
            y += c1;
            z += c2;
        }
        if (xf.flags & MVX_XF_TRANSLATE) { // `coords += translation` inside the rotation branch
            x += t0;
            y += t1;
            z += t2;
        }
    }
    if (xf.flags & MVX_XF_TRANSLATE) { // ... and `coords = coords + translation` again (reference quirk Q4)
        x = x + t0;
        y = y + t1;
        z = z + t2;
    }
}

// The same transform in float32, for the per-molecule kernel's candidate scan only: a cheap estimate of where the atom lands,
// p' = M (p - c) + o with M the matrix of q p conj(q) (identity without a rotation). Every float32 operation is off by
// at most 2^-24 of its result and all intermediates are bounded by s (|p|_1 + |c|_1) + |o|_1, s = max(1, |q|^2), on
// paths a handful of operations deep: the estimate is within ~1e-6 of that magnitude of the float64 result. The scan
// widens every test by SCAN_MARGIN (2e-5) times the magnitude, so its candidate set stays a superset; membership is
// decided later in float64 (prep_atom / the stage step), never here.
constexpr float SCAN_MARGIN = 2.0e-5f;
// Workgroup-uniform values that the vector ALU computed (there is no scalar float arithmetic) are moved to scalar
// registers explicitly: left in VGPRs they are the first thing the allocator spills, and one scratch reload inside
// a dependent chain costs a memory round trip.
__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ double uniform(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
struct XformF32 {
    float c0, c1, c2;                                  // subtracted first
    float m00, m01, m02, m10, m11, m12, m20, m21, m22; // rotation (only when rot)
    float o0, o1, o2;                                  // added last
    float scale;                                       // max(1, |q|^2)
    float mag;                                         // scale * |c|_1 + |o|_1 + 1
    bool rot;
};
__device__ __forceinline__ XformF32 make_xform_f32(const mvx_xform &xf) {
    XformF32 X;
    double c0 = xf.center[0], c1 = xf.center[1], c2 = xf.center[2];
    if (xf.flags & MVX_XF_CENTER_PTR) {
        c0 = xf.center_ptr[0];
        c1 = xf.center_ptr[1];
        c2 = xf.center_ptr[2];
    }
    const bool cen = (xf.flags & MVX_XF_CENTER) != 0, rot = (xf.flags & MVX_XF_ROTATE) != 0;
    const bool tr = (xf.flags & MVX_XF_TRANSLATE) != 0, rec = rot && (xf.flags & MVX_XF_RECENTER) != 0;
    X.rot = rot;
    X.c0 = cen ? (float)c0 : 0.0f;
    X.c1 = cen ? (float)c1 : 0.0f;
    X.c2 = cen ? (float)c2 : 0.0f;
    const float tm = tr ? (rot ? 2.0f : 1.0f) : 0.0f; // the translation is applied twice after a rotation (quirk Q4)
    X.o0 = tm * xf.trans[0] + (rec ? (float)c0 : 0.0f);
    X.o1 = tm * xf.trans[1] + (rec ? (float)c1 : 0.0f);
    X.o2 = tm * xf.trans[2] + (rec ? (float)c2 : 0.0f);
    const float q0 = (float)xf.quat[0], q1 = (float)xf.quat[1], q2 = (float)xf.quat[2], q3 = (float)xf.quat[3];
    X.m00 = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
    X.m01 = 2.0f * (q1 * q2 - q0 * q3);
    X.m02 = 2.0f * (q1 * q3 + q0 * q2);
    X.m10 = 2.0f * (q1 * q2 + q0 * q3);
    X.m11 = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
    X.m12 = 2.0f * (q2 * q3 - q0 * q1);
    X.m20 = 2.0f * (q1 * q3 - q0 * q2);
    X.m21 = 2.0f * (q2 * q3 + q0 * q1);
    X.m22 = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
    const float n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    X.scale = rot ? (n2 > 1.0f ? n2 : 1.0f) * 1.001f : 1.0f;
    X.mag = X.scale * (fabsf(X.c0) + fabsf(X.c1) + fabsf(X.c2)) + fabsf(X.o0) + fabsf(X.o1) + fabsf(X.o2) + 1.0f;
    float *fields[] = {&X.c0, &X.c1, &X.c2, &X.m00, &X.m01, &X.m02, &X.m10, &X.m11, &X.m12, &X.m20, &X.m21, &X.m22,
                       &X.o0, &X.o1, &X.o2, &X.scale, &X.mag};
    for (float *fp : fields) *fp = uniform(*fp);
    return X;
}

__device__ __forceinline__ int find_molecule(const int64_t *offsets, int B, int64_t a) {
    int lo = 0, hi = B; // offsets[lo] <= a < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= a) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// per-atom preparation: culls -> admitted voxel ranges, threshold, coefficient
// ------------------------------------------------------------------------------------------------
// Admitted reference-block interval along one axis, as voxel indices (numpy/voxelizer.py:500-513):
// block b admits the atom iff (b == 0 or p > bounds[b-1] - r) and (b == nb-1 or p < bounds[b] + r),
// bounds[m] = axis[(m+1)*bd] + res/2 (numpy/voxelizer.py:55). Both conditions are monotone in b, so
// the admitted set is the interval [#(p >= bounds[m] + r), #(p > bounds[m] - r)].
// Each count is the index where its (monotone) predicate flips, so an estimate from one division is walked to the
// flip with the reference's own comparisons: exact whatever the estimate was, one or two comparisons instead of nb-1.
__device__ __forceinline__ void block_interval(const Geom &g, double p, double r, int &vlo, int &vhi) {
    const double hres = g.res / 2.0;
    const int last = g.nb - 1; // counts range over [0, nb-1]
    auto bound = [&](int m) { return ((double)((m + 1) * g.bd) * g.res - g.half) + hres; }; // numpy/voxelizer.py:55
    const double inv_pitch = g.inv_pitch; // 1 / (bd * res): the estimates below need not be exact
    auto clampi = [&](double v) { return v < 0.0 ? 0 : (v > (double)last ? last : (int)v); };
    // count = #{m < nb-1 : cond(m)} for a predicate that is true exactly below the count: the estimate is right when
    // cond fails at it and holds just below it - two evaluations, straight-line - and is walked to the flip otherwise
    // (rare; the loops must not be interleaved: left to itself the compiler evaluates four bounds per trip, ~120
    // instructions before a loop can leave, 840 of the kernel's 890 vector instructions per atom).
    auto settle = [&](int b, auto cond) {
        const bool up = (b < last) & cond(b), down = (b > 0) & !cond(b - 1);
        if (up | down) {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (b < last && cond(b)) ++b;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (b > 0 && !cond(b - 1)) --b;
        }
        return b;
    };
    // bhi = #{m < nb-1 : p > bound(m) - r}
    const int bhi = settle(clampi(floor((p + r + g.half - hres) * inv_pitch)), [&](int m) { return p > bound(m) - r; });
    // blo = #{m < nb-1 : !(p < bound(m) + r)}
    const int blo = settle(clampi(floor((p - r + g.half - hres) * inv_pitch)), [&](int m) { return !(p < bound(m) + r); });
    vlo = blo * g.bd;
    vhi = (bhi + 1) * g.bd - 1;
    if (vhi > g.D - 1) vhi = g.D - 1;
}

// lo = 0xffff, hi = 0: fails every overlap test (lo <= box_hi needs box_hi >= 65535, beyond any grid)
constexpr uint32_t EMPTY_RANGE = 0x0000ffffu;
constexpr uint32_t EMPTY_ENTRY = 0x00ff00ffu; // packed y/z slab ranges: y lo = z lo = 255, hi = 0: matches no slab

// Everything the path knows about one atom once its position p (after centring / transform) is fixed: the culls of
// rule steps 1-2 folded into admitted voxel ranges, the membership threshold T, the gaussian coefficient k. Shared by
// prep_kernel (one thread per atom, records to memory) and voxelize_pair_kernel's per-lane-range stage (one lane per candidate, records
// straight into LDS). rmax32 / rmax64: max channel radius (RAD_CHANNEL_FEATURES only). Returns false when no voxel
// can receive a contribution (ranges then are EMPTY_RANGE).
__device__ __forceinline__ bool prep_atom(const PrepArgs &A, int64_t a, const double (&p)[3], float rmax32, double rmax64,
                                          AtomRec &R, uint32_t (&rng)[3]) {
    const bool f64 = (A.precision == 64);

    const Geom g = A.g;
    const double ub = g.half, lb = -1 * g.half;
    float r32;   // membership radius (float32, as np.divide sees it)
    double rc;   // fp64 radius the culls use
    double rwin; // widest radius for the conservative index window
    double r64 = 0.0; // float64 grids: the membership radius as np.divide sees it
    bool keep = true;
    int32_t type = 0;
    if (A.types) {
        type = A.types[a];
        if (type < 0 || type >= A.C) keep = false; // never index radii / channels out of range
    }
    if (A.radii_src == RAD_SCALAR) {
        rc = A.radius_scalar;
        r32 = (float)A.radius_scalar;
        r64 = A.radius_scalar;
        rwin = f64 ? r64 : (double)r32;
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lb - rc) && (p[i] < ub + rc); // numpy/voxelizer.py:487-488
    } else if (A.radii_src == RAD_CHANNEL_FEATURES) {
        double lo, hi;
        if (f64) { // np.float64 scalar: plain float64 arithmetic
            r64 = rmax64;
            r32 = (float)r64;
            rc = rwin = r64;
            lo = lb - r64;
            hi = ub + r64;
        } else {
            const float rmax = rmax32;
            r32 = rmax;
            rc = rwin = (double)rmax;
            // np.float32 scalar: (python float -/+ float32) is evaluated in float32 (NEP 50), numpy/voxelizer.py:138
            lo = (double)((float)lb - rmax);
            hi = (double)((float)ub + rmax);
        }
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lo) && (p[i] < hi);
    } else {
        const int64_t ri = (A.radii_src == RAD_ATOM) ? a : (keep ? (int64_t)type : -1); // numpy/voxelizer.py:284-285
        if (f64) {
            r64 = ri >= 0 ? static_cast<const double *>(A.radii)[ri] : 0.0;
            r32 = (float)r64;
            rc = rwin = r64;
        } else {
            r32 = ri >= 0 ? static_cast<const float *>(A.radii)[ri] : 0.0f;
            rc = rwin = (double)r32;
        }
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] + rc > lb) && (p[i] - rc < ub); // numpy/voxelizer.py:491-492
    }

    R.px = p[0];
    R.py = p[1];
    R.pz = p[2];
    if (f64) { // float64 grids: threshold on d2 in the T slot, the float64 gaussian coefficient in the last two pad words
        R.T = d2_threshold64(r64);
        R.k = 0.0f;
    } else if (A.radii_src == RAD_SCALAR) { // one radius for every atom: evaluated once, on the host (same IEEE operations)
        R.T = A.T_scalar;
        R.k = A.k_scalar;
    } else {
        R.T = d2_threshold(r32);
        R.k = (A.density == MVX_GAUSSIAN) ? gauss_coeff(r32, A.sigma32) : 0.0f;
    }
    R.type = type;
    R.pad[0] = R.pad[1] = R.pad[2] = 0;
    if (f64 && A.density == MVX_GAUSSIAN && R.T >= 0.0) {
        const double c64 = gauss_coeff64(r64, A.sigma64);
        __builtin_memcpy(&R.pad[1], &c64, 8); // (byte 56 of the record: 8-byte aligned)
    }
    keep = keep && (R.T >= 0.0);

    rng[0] = rng[1] = rng[2] = EMPTY_RANGE;
    if (keep) {
        // Voxels that can pass |p - g_i| <= r are i in [ceil((p - r - g0)/res), floor((p + r - g0)/res)]; the
        // radius is widened by 1e-6 relative (fp64 rounding of this estimate is ~1e-15) so the window is a
        // superset of the membership set; membership itself is decided per voxel with the exact threshold.
        // (a multiplication by 1/res is off by ~1e-13 voxels here, the widening is >= 1e-9: still a superset)
        const double rr = rwin * 1.000001 + 1e-9;
        const double inv_res = g.inv_res; // 1.0 / res, rounded once on the host
        for (int i = 0; i < 3; ++i) {
            double flo = ceil((p[i] - rr + g.half) * inv_res);
            double fhi = floor((p[i] + rr + g.half) * inv_res);
            flo = flo < 0.0 ? 0.0 : flo;
            fhi = fhi > (double)(g.D - 1) ? (double)(g.D - 1) : fhi;
            if (!(flo <= fhi)) {
                keep = false;
                break;
            }
            int lo = (int)flo, hi = (int)fhi;
            if (g.nb > 1) { // exact reference-block cull
                int vlo, vhi;
                block_interval(g, p[i], rc, vlo, vhi);
                lo = lo > vlo ? lo : vlo;
                hi = hi < vhi ? hi : vhi;
            }
            if (lo > hi) {
                keep = false;
                break;
            }
            rng[i] = (uint32_t)lo | ((uint32_t)hi << 16);
        }
    }
    if (!keep) rng[0] = rng[1] = rng[2] = EMPTY_RANGE;
    R.xr = rng[0];
    R.yr = rng[1];
    R.zr = rng[2];
    return keep;
}

// ---- per-molecule kernel (mvx_pair.hip): block cull of ONE reference block, uniform over a slab ----
// reference block cull along one axis for the block holding voxel index v (numpy/voxelizer.py:500-513):
// lo / hi are bounds[b-1] and bounds[b] (numpy/voxelizer.py:55), has_lo / has_hi say whether the comparison applies
struct BlockBounds {
    double lo, hi;
    bool has_lo, has_hi;
};
__device__ __forceinline__ BlockBounds block_bounds(const Geom &g, int v) {
    BlockBounds B;
    int blk = v / g.bd;
    if (blk > g.nb - 1) blk = g.nb - 1;
    const double hres = g.res / 2.0;
    B.has_lo = g.nb > 1 && blk >= 1;
    B.has_hi = g.nb > 1 && blk <= g.nb - 2;
    B.lo = uniform(((double)(blk * g.bd) * g.res - g.half) + hres);       // bounds[blk - 1]
    B.hi = uniform(((double)((blk + 1) * g.bd) * g.res - g.half) + hres); // bounds[blk]
    return B;
}
// the same without the integer division and without moving anything to scalar registers (voxelize_pair_kernel: evaluated
// once, by the few lanes that prepare records)
__device__ __forceinline__ BlockBounds block_bounds_lane(const Geom &g, int v) {
    BlockBounds B;
    int blk = g.bd > 1 ? (int)__umulhi((unsigned)v, g.bd_inv) : v;
    if (blk > g.nb - 1) blk = g.nb - 1;
    const double hres = g.res / 2.0;
    B.has_lo = g.nb > 1 && blk >= 1;
    B.has_hi = g.nb > 1 && blk <= g.nb - 2;
    B.lo = ((double)(blk * g.bd) * g.res - g.half) + hres;       // bounds[blk - 1]
    B.hi = ((double)((blk + 1) * g.bd) * g.res - g.half) + hres; // bounds[blk]
    return B;
}
__device__ __forceinline__ bool block_admits(const BlockBounds &B, double p, double r) {
    return (!B.has_lo || p > B.lo - r) && (!B.has_hi || p < B.hi + r);
}

// candidate lists written by xbin_kernel (mvx_prep.hip) and read by the slab kernels
constexpr int XL_HEADER = 2;
constexpr int XL_LDS = 1024; // x-list entries cached in LDS for pass B (8 KB; + 8 KB of lines: 8 blocks per CU); longer lists are re-read from L2
constexpr int SLOTS = 64;      // primary slab line: header + 63 candidates = 512 B, one per slab, densely packed
constexpr int EXT_SLOTS = 192; // extension line (entries 64..255) in a separate array: touched only by dense slabs
constexpr int LINE_CAP = SLOTS + EXT_SLOTS - 1; // candidates a slab can hold before it takes the x-list path
constexpr unsigned LINE_OVERFLOW = 0xffffffffu;
static_assert(SLOTS == SLAB_LINE_ENTRIES && EXT_SLOTS == SLAB_EXT_ENTRIES, "slab line sizes are shared with the host side");

// ------------------------------------------------------------------------------------------------
// voxelize
// ------------------------------------------------------------------------------------------------
// Shared decomposition: one slab = SUBX x SUBY x (SUBZ*NW) voxels = NW waves, one 64-voxel sub-tile per wave, one
// voxel per lane, CT channel accumulators per lane in registers. Every output byte is written exactly once (zeros
// included) with 16-B/lane non-temporal stores in whole-row runs; no atomics, no memset.
//
// voxelize_kernel (built for the normal case: the slab's primary line holds all its candidates, <= 63).
//   grid = (slab id, molecule * ncc + channel chunk), one workgroup per slab:
//     1. every wave reads the line's header {count, first atom} and the atom indices of the <= 8 row slots it stages
//        through the scalar path;
//     2. stage: wave w copies the rows of candidates w, w+NW, ... (64-B record + CT channel weights, one coalesced
//        load each, all loads in flight at once) into LDS; one barrier;
//     3. walk: each wave picks the candidates that can reach its sub-tile from the staged records (one lane per row:
//        z range, exact sphere / box cull; ballot) and processes them: fp64 d2 in cdist order, compare with T, exp2,
//        then the channel update - vector ALU: broadcast LDS reads of the weight row, software-pipelined against packed
//        FMAs; 32-channel chunks: two candidates per v_mfma_f32_32x32x2_f32 pair, two voxels per lane (OpsMx32);
//     4. write-out: accumulators -> LDS tile (4 channels per round on the vector path, 8 on the matrix path) -> stores.
//        Empty slabs skip the LDS round trip.
//   A slab with 64..255 candidates repeats 1-3 over the rest of the line and its extension (rounds of 64 rows, the
//   accumulators carried along); a slab beyond that (LINE_OVERFLOW) does the same over its (molecule, x-slab) list.
//   Channel-wise radii for features: the GROUPED instantiation (one threshold / density per distinct radius and candidate);
//   the CHANWISE instantiation (one per channel) only when there are more than 32 distinct radii.
// voxelize64_kernel: the same slab body for float64 grids with chunks of 32 channels (OpsMx64, v_mfma_f64_16x16x4_f64).
// voxelize_dense_kernel (the general slab loop, float64 grids of <= 16 channels or with channel-wise radii: grid-stride
//   over all slabs). Per slab: rounds of 64 entries over the primary + extension line (<= 255 candidates), or, beyond
//   that, wave 0 compacts the (molecule, x-slab) list in rounds of LCAP entries into an LDS list and the rows are staged
//   in rounds of dcap; same walk and write-out.
// LDS map (dynamic, 16-B aligned): voxelize_kernel: union { 64 x SW words of rows ; (CR*RPC rows) x RS floats tile };
//   dense kernel: int list[LCAP] | uint32 zr[LCAP] | int nlist | union { dcap rows ; tile }, LCAP = 64 * min(NW, 4).

// 16-B output store: non-temporal. Output bytes are written once and never re-read here; nt keeps them from displacing
// the re-read inputs in L2 (0.69 -> 0.54 ms, cfg-2; sc1 = plain).
__device__ __forceinline__ void store_f4(float *dst, const float4 v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f4 *>(dst));
}

// floats per tile row: SUBZ*NW plus a pad that keeps ds_write_b32 conflict-free for the lane -> (row, column) map
__host__ __device__ __forceinline__ int row_stride_floats(int NW) { return SUBZ * NW + 8; }
// words per staged row: 16 of record + the channel weights, padded to an ODD number of 16-B quads - the row filter reads one
// row per lane (ds_read_b128 at a lane stride of one row): with 12 quads per row (CT = 32) sixteen lanes fell on four
// distinct bank groups (4-way conflict, and 16-way for the 4-byte read of the z range); with 13 they are conflict-free
__host__ __device__ constexpr int cand_stride_words(int ct) {
    const int w = 16 + (ct < 4 ? 4 : ct);
    return ((w / 4) & 1) ? w : w + 4;
}

// what a lane knows about its voxel and its workgroup's slab
struct LaneCtx {
    double gx, gy, gz; // voxel centre: axis[i] = i*res - width/2 (numpy/voxelizer.py:41-43)
    double gx1;        // OpsMx32 / OpsMx64 (several voxels per lane): the x + 1 plane's coordinate
    double gy1;        // OpsMx64 only (four voxels per lane: x, x + 1 times iy, iy + 2): the second y row's coordinate
    int grp;           // grouped launches (channel-wise features by radius) only: the radius slot of channel cbase + lane % 32
                       // (-1: no such channel), the number of slots of this chunk (uniform) and the LDS copy of the
    int nslots;        // slots' {T, k}, by descending radius (float64 matrix-core path: the 2^(j/64) table)
    const double *gtab;
    int ix, iy, iz;
    int zt_w;          // this wave's sub-tile index along z
    int cbase;         // first channel of this workgroup's chunk
};

// One candidate (row r staged in LDS) into the accumulators.
template <int CT, bool GAUSS, bool LANE_RANGE>
__device__ __forceinline__ void accumulate_row(float2v (&acc)[(CT + 1) / 2], const unsigned *r, const LaneCtx &L) {
    const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
    const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
    uint4 q;                                                       // k, type, xr, yr
    if (LANE_RANGE) q = *reinterpret_cast<const uint4 *>(r + 8);
    else q.x = r[8]; // (k alone: a 4-byte broadcast read is half the LDS cycles of a 16-byte one)
    const double dx = Pxy.x - L.gx, dy = Pxy.y - L.gy, dz = PzT.x - L.gz;
    const double d2 = (dx * dx + dy * dy) + dz * dz; // cdist order, no fma
    bool hit = d2 <= PzT.y;
    if (LANE_RANGE) {
        const unsigned zr = r[12];
        hit = hit && (L.ix >= (int)(q.z & 0xffff)) && (L.ix <= (int)(q.z >> 16)) && (L.iy >= (int)(q.w & 0xffff)) &&
              (L.iy <= (int)(q.w >> 16)) && (L.iz >= (int)(zr & 0xffff)) && (L.iz <= (int)(zr >> 16));
    }
    const float d2f = (float)d2;
    const float *f = reinterpret_cast<const float *>(r + 16);
    // no early-out on "no lane hit": ~85 % of the filtered candidates hit, and a straight-line body pipelines
    const float ev = GAUSS ? __builtin_amdgcn_exp2f(__uint_as_float(q.x) * d2f) : 1.0f;
    const float val = hit ? ev : 0.0f;
    if constexpr (CT == 1) {
        acc[0].x = fmaf(val, f[0], acc[0].x);
    } else {
        const float2v v2 = (float2v){val, val};
        if constexpr (CT >= 16) {
            // software pipeline over the weight row: two 16-B LDS reads in flight while the packed FMAs of the
            // previous pair issue (left to itself hipcc serialises read -> wait -> 2 FMAs eight times)
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v *f4 = reinterpret_cast<const f4v *>(f);
            f4v A = f4[0], B = f4[1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < CT / 4; g += 2) {
                acc[2 * g + 0] = __builtin_elementwise_fma(v2, (float2v){A.x, A.y}, acc[2 * g + 0]);
                acc[2 * g + 1] = __builtin_elementwise_fma(v2, (float2v){A.z, A.w}, acc[2 * g + 1]);
                if (g + 2 < CT / 4) A = f4[g + 2];
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * g + 2] = __builtin_elementwise_fma(v2, (float2v){B.x, B.y}, acc[2 * g + 2]);
                acc[2 * g + 3] = __builtin_elementwise_fma(v2, (float2v){B.z, B.w}, acc[2 * g + 3]);
                if (g + 3 < CT / 4) B = f4[g + 3];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < CT / 2; ++c) {
                const float2v f2 = *reinterpret_cast<const float2v *>(f + 2 * c);
                acc[c] = __builtin_elementwise_fma(v2, f2, acc[c]);
            }
        }
    }
}

// Write-out for grids whose rows are not whole 16-byte quads (D % 4 != 0, or a grid that is not 16-B aligned): the
// float4-per-(row, z quad) stores below would be four 4-byte stores per lane at a 16-byte lane stride. Instead: a
// (channel, x) plane's part of the slab is ONE contiguous run of the grid when the slab spans whole rows (nzc == 1: SUBY
// rows of D floats = 800 B at D = 50), else one run per row segment. A run is written as 16-byte stores from its first
// 16-B aligned float on, plus its <= 3 + 3 edge floats as 4-byte stores by other threads. The transposition tile holds
// the runs as they lie in memory: run (c, x, y) at L0 + c * SC + x * SX + y * SY with the strides congruent mod 4 to
// the grid's (D^3, D^2, D) and L0 to the first run's offset, so a 16-B aligned quad of the grid is a 16-B aligned quad of
// the tile (one ds_read_b128 per store; reading four floats at a 16-byte lane stride is an 8-way bank conflict and cost
// 20 % of the call). The strides never exceed the float4 layout's (RPC * RS, SUBY * RS, RS): same LDS allocation.
// Plain stores, not non-temporal ones: the cache lines at both ends of a run are shared with the neighbouring slab's run,
// and a line that stays in L2 until its second writer arrives goes to memory once, whole (VoxParams::xcd_ranges puts
// the two writers behind the same L2). Measured, 64 molecules per call, TB/s of grid bytes (tools/odd_d_probe.py): D = 49
// 2.18 -> 2.87, D = 63 2.76 -> 3.50, D = 64 on a grid 4 bytes off alignment 3.02 -> 3.62, D = 65 1.20 -> 2.32, D = 101
// (C = 8) 1.01 -> 1.55 (4-byte stores at a 16-byte lane stride before). One 4-byte store per lane over consecutive floats
// (no alignment cases at all) is slower than either: 1.6-2.0.
struct RunLayout {
    int SC, SX, SY; // tile floats between channels, x planes, rows
    int joined;     // the slab spans whole rows: the rows of one (channel, x) follow each other in the grid (SY = D)
    int run_len;    // floats per run
    int ny, nx;     // rows / planes of the slab inside the grid
    int seg;        // floats of a row inside the slab
};
__device__ __forceinline__ RunLayout run_layout(int NW, int x0, int y0, int z0, const VoxParams &P) {
    RunLayout R;
    const int D = P.D;
    const unsigned D2 = (unsigned)D * (unsigned)D; // (low bits only are used)
    R.joined = P.nzc == 1;
    R.ny = min(SUBY, D - y0);
    R.nx = min(SUBX, D - x0);
    R.seg = R.joined ? D : min(SUBZ * NW, D - z0);
    R.run_len = R.joined ? R.ny * D : R.seg;
    R.SY = R.joined ? D : SUBZ * NW + ((D - SUBZ * NW) & 3);
    R.SX = SUBY * R.SY + ((int)(D2 - (unsigned)(SUBY * R.SY)) & 3);
    R.SC = ((SUBX * R.SX + 3) & ~3) + (int)((D2 * (unsigned)D) & 3u);
    return R;
}
// offset of the tile's first run: congruent mod 4 to the 4-byte index of the run's first float in memory
__device__ __forceinline__ int run_tile_origin(size_t S0, const float *out) {
    return (int)(((unsigned)S0 + (unsigned)(reinterpret_cast<uintptr_t>(out) >> 2)) & 3u);
}

// `nch` tile channels starting at grid channel ch0 (S0: the first run's first float, floats from `out`); ZERO: zeros, no tile
template <bool ZERO>
__device__ __forceinline__ void store_runs(const float *tile, const RunLayout &R, int L0, int nch, int ch0, size_t S0, int tid,
                                           int nthr, float *out, const VoxParams &P) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int D = P.D, run_len = R.run_len;
    const int ry_sh = R.joined ? 0 : SUBY_SH; // runs per (channel, x): 1 | SUBY (rows beyond the grid are skipped)
    const int nruns = (min(nch, P.C - ch0) << SUBX_SH) << ry_sh;
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    // run r = ((c * SUBX + x) << ry_sh) + yy: its first float in the grid and in the tile
    auto locate = [&](int r, size_t &S, int &lbase) -> bool {
        const int yy = r & ((1 << ry_sh) - 1), cx = r >> ry_sh, x = cx & (SUBX - 1), c = cx >> SUBX_SH;
        S = S0 + (size_t)(unsigned)c * D3 + (size_t)((unsigned)x * (unsigned)D2 + (unsigned)(yy * D));
        lbase = L0 + c * R.SC + x * R.SX + yy * R.SY;
        return x < R.nx && yy < R.ny;
    };
    // 16-byte slots: thread -> (slot j of run rfirst, rfirst + rstep, ...), slots per run rounded up to a power of two
    const int QS = run_len >> 2; // a run has QS or QS - 1 whole aligned quads
    if (QS) {
        const int qsh = 32 - __builtin_clz((unsigned)QS - 1u | 1u) - (QS == 1 ? 1 : 0); // ceil(log2(QS)); 2^qsh <= nthr
        const int j = tid & ((1 << qsh) - 1), rstep = nthr >> qsh;
        for (int r = tid >> qsh; r < nruns; r += rstep) {
            size_t S;
            int lbase;
            const bool ok = locate(r, S, lbase);
            const int i0 = ((4 - lbase) & 3) + 4 * j; // (floats before the run's first aligned one) + 4 j
            if (ok && i0 + 4 <= run_len) {
                f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (!ZERO) v = *reinterpret_cast<const f4 *>(tile + lbase + i0);
                *reinterpret_cast<f4 *>(out + S + i0) = v;
            }
        }
    }
    // edge floats: thread e takes float k = e % 8 (< 6) of run e / 8 - k < 3: before the first aligned float; else after the
    // last whole quad
    for (int e = tid; e < nruns * 8; e += nthr) {
        const int r = e >> 3, k = e & 7;
        size_t S;
        int lbase;
        const bool ok = locate(r, S, lbase);
        const int a = min((4 - lbase) & 3, run_len);
        const int nfull = (run_len - a) >> 2;
        const int i = k < 3 ? k : a + 4 * nfull + (k - 3);
        if (ok && k < 6 && (k < 3 ? k < a : i < run_len)) {
            const float v = ZERO ? 0.0f : tile[lbase + i];
            out[S + i] = v;
        }
    }
}

// Write-out of one slab. `any` false: zero fill without the LDS round trip. Begins with a barrier (the union region
// may still hold candidate rows) and ends without one.
// RUNS: the kernel also serves grids whose rows are not whole 16-byte quads (store_runs). Compiled into the per-lane-range
// kernels and the run-wise kernels (voxelize_runs_kernel, voxelize_pair_runs_kernel) only: the aligned-grid kernels keep their
// register budget.
template <int CT, bool RUNS, int CRMAX = CR_F32>
__device__ __forceinline__ void write_slab(const float2v (&acc)[(CT + 1) / 2], bool any, float *tile, int tid, int lane,
                                           int wave, int NW, int b, int cbase, int x0, int y0, int z0, float *out,
                                           const VoxParams &P) {
    constexpr int CR = CT < CRMAX ? CT : CRMAX; // channels per write-out round
    constexpr int NROUND = CT / CR;
    const int D = P.D;
    const int RS = row_stride_floats(NW);
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    const int F4 = (SUBZ / 4) * NW; // float4 slots per row
    // row of this thread in pass 0 (rows advance by 4 channels = 4*RPC rows per pass) and its float4 slot inside the row.
    // (tid / F4 by a float reciprocal: (tid + 0.5) / F4 lies at least 1 / 64 away from every integer for tid < 1024, F4 <= 32 -
    // the integer division by a run-time value is ~40 instructions per wave)
    const int rfirst = (int)(((float)tid + 0.5f) * __frcp_rn((float)F4));
    const int q = tid - rfirst * F4;
    const int zq = z0 + 4 * q;
    const int sxx = (rfirst >> SUBY_SH) & (SUBX - 1), syy = rfirst & (SUBY - 1), cfirst = rfirst / RPC;
    const bool vox_ok = (x0 + sxx < D) && (y0 + syy < D) && (zq < D);
    float *dst0 = out + ((size_t)b * P.C + cbase + cfirst) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
    if (!any) {
        // Pacing: a workgroup that has nothing to compute would fire its 64 KB of stores the moment it starts; holding
        // them back ~1.7 us (4096 cycles) lets the store streams of the resident workgroups interleave: ligand batches
        // 6.2 -> 6.7 TB/s (sleep 16 / 32 / 48 / 64 / 80 / 100: +0.9 / 2.8 / 4.7 / 7.5 / 7.0 / 3.7 %).
        // (only when several rounds of workgroups follow each other; a small launch would just start later)
        if (P.pace) __builtin_amdgcn_s_sleep(EMPTY_HOLD);
        if (RUNS && !P.vec_store) {
            const RunLayout R = run_layout(NW, x0, y0, z0, P);
            const size_t S0 = (((size_t)b * P.C + cbase) * D + x0) * D2 + (size_t)y0 * D + z0;
            store_runs<true>(nullptr, R, run_tile_origin(S0, out), CT, cbase, S0, tid, NW * 64, out, P);
            return;
        }
        if (vox_ok) {
#pragma unroll
            for (int p = 0; p < (CT + 3) / 4; ++p) {
                const int c = cfirst + 4 * p;
                if (c < CT && cbase + c < P.C) store_f4(dst0 + (size_t)(4 * p) * D3, make_float4(0.f, 0.f, 0.f, 0.f));
                // ... and the fill itself goes out in pieces of two store instructions (16 KB per workgroup) ~1300 cycles
                // apart, like the write-out rounds of OpsMx32::write: ligand batches 6.55 -> 6.83 TB/s (0.85 of peak;
                // 512 / 1024 / 1536 / 2048 cycles: +2 / +3.5 / +4.3 / +3.6 %; with a first wait of 2048 instead of 4096
                // cycles: -1 / +1 %)
                if (P.pace && (p & 1) && p + 1 < (CT + 3) / 4) __builtin_amdgcn_s_sleep(EMPTY_SPLIT);
            }
        }
        return;
    }
    const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1), lx = lane >> (SUBZ_SH + SUBY_SH);
    const int col = SUBZ * wave + lz;
    const int rxy = lx * SUBY + ly;
    if (RUNS && !P.vec_store) { // rows that are not whole 16-byte quads: the tile holds the slab's runs as they lie in memory (store_runs)
        const RunLayout R = run_layout(NW, x0, y0, z0, P);
        const size_t S0 = (((size_t)b * P.C + cbase) * D + x0) * D2 + (size_t)y0 * D + z0;
        const bool zok = !R.joined || col < D; // (packed rows: a voxel beyond the row would land in the next row)
        const int mine = lx * R.SX + ly * R.SY + col;
#pragma unroll
        for (int rd = 0; rd < NROUND; ++rd) {
            const size_t S0r = S0 + (size_t)(rd * CR) * D3;
            const int L0 = run_tile_origin(S0r, out);
            __syncthreads();
            if (zok) {
#pragma unroll
                for (int c = 0; c < CR; ++c) {
                    const int cg = rd * CR + c;
                    tile[L0 + c * R.SC + mine] = (cg & 1) ? acc[cg / 2].y : acc[cg / 2].x;
                }
            }
            __syncthreads();
            store_runs<false>(tile, R, L0, CR, cbase + rd * CR, S0r, tid, NW * 64, out, P);
        }
        return;
    }
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        __syncthreads(); // candidate rows (round 0) / previous tile (later rounds) fully consumed
        if (rd == 0) VK_STAMP(4); // every wave's walk is done
        if (rd == 1) VK_STAMP(5); // round 0 transposed and its stores issued
#pragma unroll
        for (int c = 0; c < CR; ++c) {
            const int cg = rd * CR + c;
            const float v = (cg & 1) ? acc[cg / 2].y : acc[cg / 2].x;
            tile[(c * RPC + rxy) * RS + col] = v;
        }
        __syncthreads();
        if (vox_ok) {
#pragma unroll
            for (int p = 0; p < (CR + 3) / 4; ++p) {
                const int c = cfirst + 4 * p; // channel inside the round
                if (c < CR && cbase + rd * CR + c < P.C) {
                    const float4 v = *reinterpret_cast<const float4 *>(tile + (rfirst + 4 * RPC * p) * RS + 4 * q);
                    store_f4(dst0 + (size_t)(rd * CR + 4 * p) * D3, v);
                }
            }
        }
    }
}

// slab id t = zc + nzc * (sy + nsy * sx)
__device__ __forceinline__ void decode_slab(unsigned t, const VoxParams &P, int &sx, int &sy, int &zc) {
    const unsigned ty = (P.nzc == 1) ? t : __umulhi(t, P.nzc_inv); // t / nzc
    zc = (int)(t - ty * P.nzc);
    sx = (P.nsy == 1) ? (int)ty : (int)__umulhi(ty, P.nsy_inv); // ty / nsy
    sy = (int)ty - sx * P.nsy;
}

__device__ __forceinline__ LaneCtx make_lane_ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase,
                                                  const VoxParams &P) {
    const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1), lx = lane >> (SUBZ_SH + SUBY_SH);
    LaneCtx L;
    L.ix = x0 + lx;
    L.iy = y0 + ly;
    L.iz = z0 + SUBZ * wave + lz;
    L.gx = (double)L.ix * P.res - P.half;
    L.gy = (double)L.iy * P.res - P.half;
    L.gz = (double)L.iz * P.res - P.half;
    L.zt_w = zt_lo + wave;
    L.cbase = cbase;
    return L;
}

// An "Ops" type is what differs between the accumulator layouts of the slab kernels: accumulator type, staged row width,
// the per-candidate update and the write-out. OpsF32: one voxel per lane, CT float32 channels per lane on the vector ALU
// (chunks of fewer than 32 channels, per-lane-range variants); OpsMx32 / OpsPair (mvx_ops32.h): 32 channels on the
// matrix cores / candidate pairs on the vector ALU; OpsF64 / OpsMx64 (mvx_f64.hip): float64 grids.
template <int CT_, bool GAUSS, bool LANE_RANGE>
struct OpsF32 {
    static constexpr bool RUNS = LANE_RANGE; // carries the run-wise write-out (store_runs)
    static constexpr int CT = CT_;
    static constexpr bool GROUPED = false;
    static constexpr bool VSTAGE = false;
    static constexpr bool PRESTAGE = false;
    static constexpr bool CULL = true;
    typedef float2v Acc[(CT + 1) / 2];
    static constexpr int WORDS = 1;                   // 32-bit words per channel weight
    static constexpr int WW = CT;                     // weight words staged per row
    static constexpr int SW = cand_stride_words(CT);  // row stride in LDS, words
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int c = 0; c < (CT + 1) / 2; ++c) acc[c] = (float2v){0.0f, 0.0f};
    }
    static __device__ __forceinline__ void accumulate(Acc &acc, const unsigned *r, const LaneCtx &L) {
        accumulate_row<CT, GAUSS, LANE_RANGE>(acc, r, L);
    }
    static __device__ __forceinline__ LaneCtx ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase, const VoxParams &P) {
        return make_lane_ctx(lane, wave, x0, y0, z0, zt_lo, cbase, P);
    }
    static __device__ __forceinline__ void tables(LaneCtx &, char *, const VoxParams &, int) {}
    // the rows of `mask` (one bit per staged row, atom order) into the accumulators
    static __device__ __forceinline__ void walk(Acc &acc, unsigned long long mask, const unsigned *un, int lane, const LaneCtx &L,
                                                const VoxParams &, const double *__restrict__, const float *__restrict__) {
        while (mask) {
            const int sl = __builtin_ctzll(mask);
            mask &= mask - 1;
            accumulate(acc, un + sl * SW, L);
        }
    }
    static __device__ __forceinline__ void write(const Acc &acc, bool any, unsigned *un, int tid, int lane, int wave, int NW,
                                                 int b, const LaneCtx &L, int x0, int y0, int z0, void *out, const VoxParams &P) {
        write_slab<CT, LANE_RANGE>(acc, any, reinterpret_cast<float *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0,
                       static_cast<float *>(out), P);
    }
};

// Can the atom of staged row r reach ANY voxel centre of this wave's sub-tile? The candidate lists are built from index
// ranges, i.e. from the atom's bounding box: of the rows whose box meets the sub-tile's box, 15 % (radius 1 A on the 0.5 A
// grid) to 27 % (2 A) come no closer than their radius to its nearest corner, and a walked candidate costs ~116 vector
// and 44 LDS cycles whether or not a lane hits. One lane per row: distance from the atom to the box spanned by the
// sub-tile's voxel centres (centre / half-extent form, float32) against the membership threshold T. This only drops
// rows; the estimate is made a lower bound of the true distance (below), and everything it keeps is decided per voxel by
// the exact d2 <= T as before.
// Coordinates float32 cannot hold (huge or non-finite: test_non_finite_...) compare false and keep the row.
__device__ __forceinline__ bool reaches_subtile(const unsigned *r, int lane, const LaneCtx &L, const VoxParams &P) {
    const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
    const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
    const float res = (float)P.res;
    // this lane's voxel is (lx, ly, lz) inside the sub-tile: the box centre is the same for every lane
    // (lx: lanes 32..63 hold the x + 1 plane in the one-voxel-per-lane layout; in the two-voxel layout every lane's gx is
    // the x plane's and the upper half of the wave holds other candidates, not other voxels: L.ix tells which)
    const int lz = L.iz & (SUBZ - 1), ly = L.iy & (SUBY - 1), lx = L.ix & (SUBX - 1); // (sub-tile origins are multiples of its edges)
    const float cx = (float)L.gx + (0.5f * (SUBX - 1) - (float)lx) * res;
    const float cy = (float)L.gy + (0.5f * (SUBY - 1) - (float)ly) * res;
    const float cz = (float)L.gz + (0.5f * (SUBZ - 1) - (float)lz) * res;
    // every axis distance is shortened by 2e-6 of the magnitudes it was formed from (each float32 conversion and
    // operation is off by at most 6e-8 of them): the estimate never exceeds the true distance, at any grid scale
    const float px = (float)Pxy.x, py = (float)Pxy.y, pz = (float)PzT.x;
    const float hx = 0.5f * (SUBX - 1) * res, hy = 0.5f * (SUBY - 1) * res, hz = 0.5f * (SUBZ - 1) * res;
    const float ex = fmaxf(fabsf(px - cx) - hx - 2.0e-6f * (fabsf(px) + fabsf(cx) + hx), 0.0f);
    const float ey = fmaxf(fabsf(py - cy) - hy - 2.0e-6f * (fabsf(py) + fabsf(cy) + hy), 0.0f);
    const float ez = fmaxf(fabsf(pz - cz) - hz - 2.0e-6f * (fabsf(pz) + fabsf(cz) + hz), 0.0f);
    const float dmin2 = ex * ex + ey * ey + ez * ez;
    const float T = (float)PzT.y;
    return !(dmin2 > T * 1.00001f);
}

// The same round for voxelize_kernel, with the line read through the SCALAR memory path: the header and the atom
// indices of the (at most eight) slots this wave stages are wave-uniform, so they are s_load'ed (scalar cache -> L2)
// instead of travelling, 512 B per wave, through the vector memory pipeline - where a load queues behind the 64 KB of
// stores every resident workgroup pushes through the same pipeline (the line load took 3 000 cycles at the median,
// profiles/r02_phase_timelines.txt). The z sub-tile filter reads the admitted z range from the staged records instead
// of the line's packed copy (same bits: both come from prep_atom's range, in SUBZ-voxel units).
// xl (workgroup-uniform, rare): the entries come from a (molecule, x-slab) list instead of a slab line (LINE_OVERFLOW slabs,
// below): they have not been filtered against the slab's y rows, so the walk's row filter also tests the record's admitted
// y range.
// --- the rounds of voxelize_kernel: row slot sl of a round that starts at entry e0 holds entry e0 + sl of the line (entries
// 1..n_line; entry e sits at line[e] up to SLOTS-1 and at ext[e - SLOTS] beyond; lanes 0-15 of a row are the record, lanes
// 16.. the channel weights of the chunk) ---
template <typename Ops>
struct RoundSrc {
    const unsigned *src; // this lane's word of row 0 (record word / weight column)
    unsigned stride;     // words between the rows of consecutive atoms, for this lane
    bool stager;         // this lane takes part in staging
    const unsigned *src2; // rows wider than a wave (float64, 32 channels: 16 + 64 words): lanes 0.. fetch words 64.. too
    unsigned stride2;
};
template <typename Ops>
constexpr int round_tail() { return 16 + Ops::WW > 64 ? 16 + Ops::WW - 64 : 0; }
template <typename Ops>
__device__ __forceinline__ RoundSrc<Ops> round_src(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, int lane,
                                                   const LaneCtx &L, const VoxParams &P) {
    RoundSrc<Ops> R;
    R.src = lane < 16 ? rec + lane : w + (Ops::WORDS * L.cbase + lane - 16);
    R.stride = lane < 16 ? 16u : (unsigned)(Ops::WORDS * P.w_stride);
    // (grouped launches read the caller's feature rows in place whatever C is: no column beyond the row)
    R.stager = lane < 16 + Ops::WW && (!Ops::GROUPED || lane < 16 || L.cbase + lane - 16 < P.C);
    R.src2 = w + (Ops::WORDS * L.cbase + lane + 48);
    R.stride2 = (unsigned)(Ops::WORDS * P.w_stride);
    return R;
}

// Stage a round through registers: the wave's eight slots are CONSECUTIVE entries (slot 8 wave + u), so their atom indices
// arrive as one 64-byte scalar load group with no per-slot address arithmetic (slots wave + u NW - eight separate s_loads, each
// with its own 64-bit address, line / extension branch and wait - cost ~39 scalar instructions per slot, 313 per wave and
// round, on the compute unit's one scalar unit: 740 scalar against 635 vector instructions per wave at a 2.0 A radius,
// profiles/r04_staging.txt). A group never straddles line and extension (SLOTS is a multiple of 8). Row order in LDS is the
// entry order as before: the walk, and every sum, is unchanged. Then eight row loads in flight, eight LDS writes - without
// per-slot tests or exec-mask branches: a slot without a candidate (the header's, the ones past the count in the line's last
// group) takes the molecule's first row; its row in LDS is inside the 64-row region and nobody walks it. (With tests and
// branches in the groups that hold such slots - wave 0's, always - the headline launch ran at 0.805 of peak instead of 0.833,
// radius 1.5 A at 0.69 instead of 0.725, same box.) Row addresses from 32-bit operands (atom indices fit 31 bits, validate()):
// one v_mad_u64_u32 per row.
// BIG: slabs of more than 8 waves (the 1024-thread variants) - a round is 64 rows, waves 8.. stage nothing.
template <typename Ops>
__device__ __forceinline__ void stage_rows(const unsigned (&ai)[8], int s0, unsigned *un, const RoundSrc<Ops> &R, unsigned first, int lane) {
    constexpr int SW = Ops::SW;
    constexpr int TAIL = round_tail<Ops>();
    unsigned v[8], v2[TAIL ? 8 : 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        v[u] = 0u;
        if (TAIL) v2[u] = 0u;
        if (R.stager) v[u] = R.src[(size_t)(first + ai[u]) * R.stride];
        if (TAIL && lane < TAIL) v2[u] = R.src2[(size_t)(first + ai[u]) * R.stride2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = s0 + u;
        if (R.stager || (Ops::GROUPED && lane < 16 + Ops::WW)) un[sl * SW + lane] = v[u]; // (v = 0 beyond C)
        if (TAIL && lane < TAIL) un[sl * SW + 64 + lane] = v2[u];
    }
}
template <typename Ops, bool BIG>
__device__ __forceinline__ void stage_round(const uint2 *__restrict__ line, const uint2 *__restrict__ ext, int e0, int n_line,
                                            unsigned *un, const RoundSrc<Ops> &R, int64_t a0, int lane, int wave, int NW) {
    static_assert(SLOTS % 8 == 0 && EXT_SLOTS % 8 == 0, "a wave's group of eight entries lies in the line or in its extension");
    const int s0 = 8 * wave, g0 = e0 + s0; // first slot / entry of this wave (uniform)
    if ((BIG && wave >= 8) || g0 > n_line) return;
    const uint2 *__restrict__ grp = g0 < SLOTS ? line + g0 : ext + (g0 - SLOTS);
    unsigned ai[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ai[u] = grp[u].x; // (entries past n_line: inside the line / the list's slack, replaced below)
    if (g0 == 0 || g0 + 7 > n_line) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (g0 + u < 1 || g0 + u > n_line) ai[u] = 0u;
    }
    stage_rows<Ops>(ai, s0, un, R, (unsigned)a0, lane);
}

// The first TWO rounds of a slab staged at once (Ops::PRESTAGE: the 32-channel matrix-core kernels, whose row region holds
// 2 RW rows): a dense slab's second round costs a second trip through the vector memory pipeline - ~5 kcycles behind the
// compute unit's stores - plus two barriers when it is staged after the first walk (phase timelines, profiles/r04_staging.txt:
// radius 2.0 A, two to three rounds per slab: 35 kcycles per workgroup, 16-18 of them "walk"). Before the first walk the 32
// accumulator registers are not alive yet: both rounds' rows travel together (16 registers), one barrier, and the second
// round is walked straight from its own rows at un + RW * SW. Returns nothing; the caller walks round 1 when n_line >= RW.
template <typename Ops, bool BIG>
__device__ __forceinline__ void stage_first_rounds(const uint2 *__restrict__ line, const uint2 *__restrict__ ext, int n_line, int RW,
                                                   unsigned *un, const RoundSrc<Ops> &R, int64_t a0, int lane, int wave) {
    static_assert(round_tail<Ops>() == 0, "rows of at most 64 words");
    constexpr int SW = Ops::SW;
    const int s0 = 8 * wave; // first slot of this wave in either round (uniform)
    if ((BIG && wave >= 8) || s0 > n_line) return;
    const int g1 = RW + s0;
    const bool two = g1 <= n_line; // (uniform) this wave holds rows of the second round as well
    const unsigned first = (unsigned)a0;
    unsigned ai0[8], ai1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ai0[u] = line[s0 + u].x; // (s0 < 64 = SLOTS)
    if (two) {
        const uint2 *__restrict__ grp = g1 < SLOTS ? line + g1 : ext + (g1 - SLOTS);
#pragma unroll
        for (int u = 0; u < 8; ++u) ai1[u] = grp[u].x;
        if (g1 + 7 > n_line) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (g1 + u > n_line) ai1[u] = 0u;
        }
    }
    if (s0 == 0 || s0 + 7 > n_line) { // the header's slot, slots past the count: the molecule's first row, never walked
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < 1 || s0 + u > n_line) ai0[u] = 0u;
    }
    unsigned v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        v0[u] = 0u;
        if (R.stager) v0[u] = R.src[(size_t)(first + ai0[u]) * R.stride];
    }
    if (two) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v1[u] = 0u;
            if (R.stager) v1[u] = R.src[(size_t)(first + ai1[u]) * R.stride];
        }
    }
    const bool writer = R.stager || (Ops::GROUPED && lane < 16 + Ops::WW); // (v = 0 beyond C)
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (writer) un[(s0 + u) * SW + lane] = v0[u];
    if (two) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (writer) un[(g1 + u) * SW + lane] = v1[u];
    }
}

// The same round for narrow rows (16 record words + at most 16 weight words: OpsPair), with every address formed on the vector
// ALU: the round's 64 line entries arrive with ONE load (entry e0 + lane), every lane picks the atom of its row slot with
// ds_bpermute, and a load instruction fetches two rows (lanes 0-31 / 32-63). Four row loads per wave instead of eight, and
// none of stage_round's scalar work (eight s_loads, eight 64-bit scalar multiply-adds, the per-slot branches): the narrow
// launches are bound by the compute unit's one scalar unit, not - like the 32-channel kernel, which keeps the scalar path -
// by a vector memory pipeline full of stores.
template <typename Ops, bool BIG>
__device__ __forceinline__ void stage_round_v(const uint2 *__restrict__ line, const uint2 *__restrict__ ext, int e0, int n_line,
                                              unsigned *un, const unsigned *__restrict__ rec, const unsigned *__restrict__ w, int64_t a0,
                                              int lane, int wave, int NW, const LaneCtx &L, const VoxParams &P) {
    constexpr int SW = Ops::SW;
    const int e = e0 + lane;
    int ai = 0;
    if (e >= 1 && e <= n_line) ai = (int)(e < SLOTS ? line[e].x : ext[e - SLOTS].x);
    const int half = lane >> 5, wd = lane & 31;
    const bool used = wd < 16 + Ops::WW;
    const unsigned *base = wd < 16 ? rec + wd : w + (L.cbase + wd - 16);
    const unsigned stride = wd < 16 ? 16u : (unsigned)P.w_stride; // (32-bit operands: one v_mad_u64_u32 per row address)
    const unsigned first = (unsigned)a0;                          // (atom indices fit 31 bits: validate())
    // slot sl holds a candidate iff lo <= sl <= hi. Branch-free: a slot without one fetches the molecule's first row and lands in
    // a dump row behind the 64 rows of a round (the launcher allocates it) - eight exec-mask branches per wave before
    const int lo = (1 - e0) > 0 ? (1 - e0) : 0, hi = (n_line - e0) < 63 ? (n_line - e0) : 63;
    const unsigned span = (unsigned)(hi - lo);
    unsigned a[4], v[4];
    int slot[4];
    int sl = wave + NW * half; // this lane's row slot <-> entry e0 + sl
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned ar = (unsigned)__builtin_amdgcn_ds_bpermute(4 * sl, ai); // (slots beyond the round read some other lane's entry: masked)
        const bool in = (unsigned)(sl - lo) <= span;
        a[u] = in ? ar : 0u;
        slot[u] = in ? sl : 64;
        sl += 2 * NW;
        asm volatile("" : "+v"(sl)); // (one add per slot instead of a quarter-rate multiply)
    }
    if (used) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = base[(size_t)(first + a[u]) * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) un[slot[u] * SW + wd] = v[u];
    }
}

// One lane per staged row decides whether this wave walks it; then the walk. xl (workgroup-uniform, rare): the entries
// come from a (molecule, x-slab) list instead of a slab line (LINE_OVERFLOW slabs, below): they have not been filtered
// against the slab's y rows, so the row filter also tests the record's admitted y range.
template <typename Ops>
__device__ __forceinline__ void filter_walk(const bool xl, typename Ops::Acc &acc, int e0, int n_line, int RW, const unsigned *un,
                                            int lane, int wave, const LaneCtx &L, const VoxParams &P, const double *__restrict__ Tc,
                                            const float *__restrict__ kc) {
    constexpr int SW = Ops::SW;
    bool ok = false;
    if (lane < RW && e0 + lane >= 1 && e0 + lane <= n_line) {
        const unsigned *r = un + lane * SW;
        const unsigned zr = r[12];
        ok = ((int)((zr & 0xffff) >> SUBZ_SH) <= L.zt_w) && ((int)((zr >> 16) >> SUBZ_SH) >= L.zt_w);
        if constexpr (Ops::CULL) ok = ok && reaches_subtile(r, lane, L, P);
        if (xl) {
            const unsigned yr = r[11];
            const int sy = L.iy >> SUBY_SH; // (the slab's y index: the same for every lane)
            ok = ok && ((int)((yr & 0xffff) >> SUBY_SH) <= sy) && ((int)((yr >> 16) >> SUBY_SH) >= sy);
        }
    }
    Ops::walk(acc, __ballot(ok), un, lane, L, P, Tc, kc);
    VK_STAMP(8 + wave); // every wave's own walk end
}

// ------------------------------------------------------------------------------------------------
// launch helpers (host)
// ------------------------------------------------------------------------------------------------
// Dynamic LDS above the default 64 KB limit needs an opt-in per kernel and per device.
constexpr int MAX_DEVICES = 64;
struct LdsLimit {
    size_t raised[MAX_DEVICES] = {};
};

template <typename K>
static hipError_t raise_lds_limit(K kernel, size_t lds, LdsLimit &state) {
    if (lds <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (lds > state.raised[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        state.raised[dev] = lds;
    }
    return hipSuccess;
}

// Profiled launches (mvx_set_profiling): the two events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL:
// start and end timestamps of this launch, what rocprofv3 reports) instead of two hipEventRecord calls around it - an
// event recorded on the stream is a barrier packet of its own and idled the GPU ~6 us each time (kernel trace of the
// bench: 5.9 us gaps before and after every voxelize launch). timed_launch() (mvx_capi.hip) sets the pair, the next
// voxelize launch on this thread consumes it.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents &launch_events(); // (thread-local, mvx_prep.hip)
template <typename K, typename... A>
static void launch_profiled(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    LaunchEvents &ev = launch_events();
    const hipEvent_t e0 = ev.start, e1 = ev.stop;
    ev.start = ev.stop = nullptr;
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, e0, e1, 0u, args...);
    else hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
}

// blocks per molecule of the slab kernels
static inline unsigned slab_grid_x(const VoxParams &p) {
    const unsigned T = (unsigned)(p.nzc * p.nsy * p.nsx);
    if (p.xcd_ranges) return 8u * ((T + 7u) / 8u); // (mvx_slab_body.inc: every XCD a contiguous range of slabs)
    return T;
}

} // namespace mvx
