// mvx_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels of the voxelizer hot path.
//
// Replaces, on the device, the whole per-call body of the reference's
//   Voxelizer.forward_features / forward_types / forward_single
//   (molvoxel/voxelizer/numpy/voxelizer.py:97-169, 240-315, 370-436 and the helpers they call:
//    _get_overlap :481-494, _get_overlap_blocks :496-527, _calc_grid :531-560,
//    _set_grid_* :194-236, 344-366, 457-477; transform numpy/transform.py:44-60).
// It is NOT a translation of that code (Python loop over 8^3 blocks, cdist -> (V,512) -> matmul) and
// not of the torch path. Formulation: voxel-tile GATHER.
//
//   prep_kernel      one thread per atom: rigid transform in the reference's fp64 op order, exact box cull +
//                    per-axis reference-block cull folded into an admitted voxel-index range, exact membership
//                    threshold T on d2, gaussian coefficient k -> one row per atom [record | channel weights].
//   xbin_kernel      ordered (ballot/prefix, no atomics) candidate lists: per (molecule, x-slab) and, from
//                    those, one 512-B candidate line per output slab.
//   voxelize_kernel  one workgroup per output slab of 2 x 4 x (8*NW) voxels (NW waves, one 2x4x8 sub-tile per
//                    wave, one voxel per lane, CT channel accumulators per lane in registers): load the slab's
//                    candidate line, stage the candidates' rows in LDS, every wave walks the candidates that
//                    touch its sub-tile (fp64 d2, compare with T, exp2, channel update), then the accumulators are
//                    transposed through LDS and written with non-temporal 16-B/lane stores in whole-row runs.
//                    Every output byte is written exactly once, zeros included (the reference's overwrite
//                    semantics, numpy/voxelizer.py:133-135,158-160); no atomics, no memset, no (V, DHW)
//                    intermediate; HBM-write bound. The channel update of 32-channel chunks - the one dense
//                    contraction in the walk, the reference's own matmul - runs on the matrix cores in exact
//                    float32 (OpsMx32: v_mfma_f32_32x32x2_f32, bit-identical to the fmaf chain), narrower chunks on
//                    the vector ALU (packed FMAs).
//
// Exactness: membership float32(float32(sqrt_f64(d2))/r32) <= 1 is equivalent to d2 <= T with
//   y  = largest fp64 whose float32 rounding is <= r32,  T = round_down(y * nextup(y))
// (derivation in DESIGN.md §4; checked against 20k radii on the CPU and by tests/test_hip_parity.py).
// d2 is formed exactly like scipy cdist: (dx*dx + dy*dy) + dz*dz in fp64 WITHOUT fma, so this TU
// must be compiled with -ffp-contract=off and without fast-math.
#include "mvx_internal.h"

#include <math.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <cstdlib>

// cdist-order arithmetic must not be fused, whatever flags the TU is built with.
#pragma clang fp contract(off)

namespace mvx {

typedef float float2v __attribute__((ext_vector_type(2)));

#ifdef MVX_DIAG
// Diagnostic builds (tools/ab_build.sh diag "-DMVX_DIAG"): s_memtime stamps of the batched voxelize_kernel, 16 x 8 B per
// workgroup, into a buffer the host hands over with set_diag_buffer(); the shipped library has none of this.
__device__ unsigned long long *g_diag = nullptr;
#define VK_STAMP(i) do { if (g_diag && lane == 0 && (wave == 0 || (i) >= 8)) g_diag[16 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
hipError_t set_diag_buffer(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &p, sizeof(p)); }
__device__ unsigned long long *g_diag_xb = nullptr; // xbin_kernel: 8 x 8 B per block, stamps by thread 0
#define XB_STAMP(i) do { if (g_diag_xb && threadIdx.x == 0) g_diag_xb[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
hipError_t set_diag_buffer_xb(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_xb), &p, sizeof(p)); }
#else
#define VK_STAMP(i) do { } while (0)
#define XB_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// small device helpers
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
__host__ __device__ double d2_threshold(float r32) {
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
__device__ void apply_xform(const mvx_xform &xf, double &x, double &y, double &z) {
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

// The same transform in float32, for the direct kernel's candidate scan only: a cheap estimate of where the atom lands,
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
// channel-wise auxiliary: max radius (float32), per-channel thresholds / coefficients
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chan_aux_kernel(const float *radii, int C, int density, float sigma32, float *rmax, double *Tc,
                                                       float *kc, ChanGroups *groups, int *chan_slot) {
    // slot of a channel = number of distinct radii (float32 bits) that appear before its radius' first appearance
    constexpr int CMAX = 2048; // more channels: no grouping
    __shared__ unsigned rb[CMAX];
    __shared__ unsigned char is_first[CMAX];
    __shared__ int over;
    if (threadIdx.x == 0) over = 0;
    const bool small = C <= CMAX;
    for (int c = threadIdx.x; c < C && small; c += blockDim.x) rb[c] = __float_as_uint(radii[c]);
    __syncthreads();
    for (int c = threadIdx.x; c < C && small; c += blockDim.x) {
        bool f = true;
        for (int i = 0; i < c && f; ++i) f = rb[i] != rb[c];
        is_first[c] = f ? 1 : 0;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float r = radii[c];
        Tc[c] = d2_threshold(r);
        kc[c] = density == MVX_GAUSSIAN ? gauss_coeff(r, sigma32) : 0.0f;
        if (small) {
            int first = c;
            for (int i = 0; i < c; ++i)
                if (rb[i] == rb[c]) {
                    first = i;
                    break;
                }
            int slot = 0;
            for (int i = 0; i < first; ++i) slot += is_first[i];
            chan_slot[c] = slot < CHAN_GROUP_SLOTS ? slot : -1;
            if (slot >= CHAN_GROUP_SLOTS) over = 1;
            else if (first == c) {
                groups->slot[slot].T = d2_threshold(r);
                groups->slot[slot].k = density == MVX_GAUSSIAN ? gauss_coeff(r, sigma32) : 0.0f;
                groups->slot[slot].pad = 0;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = radii[0];
        for (int c = 1; c < C; ++c) m = radii[c] > m ? radii[c] : m;
        rmax[0] = m;
        int n = 0;
        for (int c = 0; c < C && small; ++c) n += is_first[c];
        groups->nslots = n;
        groups->fallback = (!small || over) ? 1 : 0;
    }
}

// float64 grids: the per-channel radii themselves (the kernel divides by them) and their maximum, in float64
__global__ void chan_aux64_kernel(const double *radii, int C, int density, double sigma, double *rmax, double *Tc, double *kc) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        Tc[c] = d2_threshold64(radii[c]);
        kc[c] = (density == MVX_GAUSSIAN && Tc[c] >= 0.0) ? gauss_coeff64(radii[c], sigma) : 0.0;
    }
    if (threadIdx.x == 0) {
        double m = radii[0];
        for (int c = 1; c < C; ++c) m = radii[c] > m ? radii[c] : m;
        rmax[0] = m;
    }
}

hipError_t launch_chan_aux64(const double *radii, int32_t C, int32_t density, double sigma, double *rmax, double *Tc, double *kc,
                             hipStream_t s) {
    hipLaunchKernelGGL(chan_aux64_kernel, dim3(1), dim3(256), 0, s, radii, C, density, sigma, rmax, Tc, kc);
    return hipGetLastError();
}

hipError_t launch_chan_aux(const float *radii, int32_t C, int32_t density, float sigma32, float *rmax, double *Tc,
                           float *kc, ChanGroups *groups, int32_t *chan_slot, hipStream_t s) {
    hipLaunchKernelGGL(chan_aux_kernel, dim3(1), dim3(256), 0, s, radii, C, density, sigma32, rmax, Tc, kc, groups, chan_slot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// prep: per-atom records
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
// prep_kernel (one thread per atom, records to memory) and voxelize_direct_kernel (one lane per candidate, records
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

// Packs the channel weights too when the voxelize kernels cannot read the caller's feature rows as they are
// (one-hot type / 1 / zero padded features): the block copies the weights of its 256 atoms cooperatively.
__global__ void __launch_bounds__(256) prep_kernel(PrepArgs A) {
    const int64_t a = A.first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool f64 = (A.precision == 64);
    if (A.wbuf) {
        const int64_t first = A.first + (int64_t)blockIdx.x * 256;
        const int nat = (int)((A.total - first) < 256 ? (A.total - first) : 256);
        for (int i = threadIdx.x; i < nat * A.Cpad; i += 256) {
            const int al = i / A.Cpad, c = i - al * A.Cpad;
            double f = 0.0;
            if (c < A.C) {
                if (A.mode == MODE_FEATURES)
                    f = f64 ? static_cast<const double *>(A.features)[(first + al) * A.C + c]
                            : (double)static_cast<const float *>(A.features)[(first + al) * A.C + c];
                else if (A.mode == MODE_TYPES) f = (A.types[first + al] == c) ? 1.0 : 0.0;
                else f = 1.0;
            }
            if (f64) static_cast<double *>(A.wbuf)[first * A.Cpad + i] = f;
            else static_cast<float *>(A.wbuf)[first * A.Cpad + i] = (float)f;
        }
    }
    // (no early return: every thread takes part in the record transposition below)
    const bool live = a < A.total;
    const int64_t al = live ? a : A.total - 1; // (A.total > A.first: launch_prep)
    double p[3] = {A.coords[3 * al], A.coords[3 * al + 1], A.coords[3 * al + 2]};
    if (A.xforms) apply_xform(A.xforms[find_molecule(A.offsets, A.B, al)], p[0], p[1], p[2]); // (8 dependent loads: only when needed)
    else if (A.xf_one.flags) apply_xform(A.xf_one, p[0], p[1], p[2]); // one molecule: its transform came with the launch
    float rmax32 = 0.0f;
    double rmax64 = 0.0;
    if (A.radii_src == RAD_CHANNEL_FEATURES) {
        if (f64) rmax64 = static_cast<const double *>(A.chan_aux)[0];
        else rmax32 = static_cast<const float *>(A.chan_aux)[0];
    }
    AtomRec R;
    uint32_t rng[3];
    const bool keep = prep_atom(A, al, p, rmax32, rmax64, R, rng);
    {   // Records leave through LDS so that a store instruction writes 1 KB of consecutive bytes: straight from the
        // registers it wrote 64 pieces of 16 B, 64 B apart (eight partial writes per 128-B line; WRITE_SIZE was 1.5 x
        // the bytes stored and the kernel store-bound). Piece i of record r sits at stage[i * PITCH + r]: writes
        // (consecutive r) and reads (16 lanes = four records x four pieces) are both bank-conflict free.
        // Non-temporal: records are not re-read by this XCD; kept out of L2 they cost the voxelize kernel 2.4 % less.
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        constexpr int PITCH = 260;
        __shared__ u4v stage[4 * PITCH];
        const u4v *src = reinterpret_cast<const u4v *>(&R);
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[i * PITCH + threadIdx.x] = src[i];
        __syncthreads();
        const int64_t first = A.first + (int64_t)blockIdx.x * 256;
        const int nrec = (int)((A.total - first) < 256 ? (A.total - first) : 256);
        u4v *dstv = reinterpret_cast<u4v *>(A.rec + first);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int slot = k * 256 + threadIdx.x, r = slot >> 2, i = slot & 3;
            if (r < nrec) __builtin_nontemporal_store(stage[i * PITCH + r], dstv + slot);
        }
    }
    // y range in SUBY-voxel slabs (lo | hi << 8), z range in SUBZ-voxel sub-tiles (lo << 16 | hi << 24); a dropped
    // atom matches no slab (EMPTY_ENTRY)
    const uint32_t packed = !keep ? EMPTY_ENTRY
                                  : ((rng[1] & 0xffff) >> SUBY_SH) | (((rng[1] >> 16) >> SUBY_SH) << 8) |
                                        (((rng[2] & 0xffff) >> SUBZ_SH) << 16) | (((rng[2] >> 16) >> SUBZ_SH) << 24);
    if (live) A.xp[a] = make_uint2(rng[0], packed);
}

hipError_t launch_prep(const PrepArgs &a, hipStream_t s) {
    if (a.total <= a.first) return hipSuccess;
    const unsigned blocks = (unsigned)((a.total - a.first + 255) / 256);
    hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) transform_kernel(const double *coords, int64_t N, const mvx_xform *xf,
                                                         double *out) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= N) return;
    double x = coords[3 * a], y = coords[3 * a + 1], z = coords[3 * a + 2];
    apply_xform(xf[0], x, y, z);
    out[3 * a] = x;
    out[3 * a + 1] = y;
    out[3 * a + 2] = z;
}

hipError_t launch_transform(const double *coords, int64_t N, const mvx_xform *xf_dev, double *out, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(transform_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, coords, N, xf_dev, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// binning: ordered x-slab lists and per-slab candidate lists
// ------------------------------------------------------------------------------------------------
// One 256-thread block per (molecule, SUBX-voxel x-slab).
//  A. x-list: the atoms whose admitted x range touches the slab, in atom order (ballot + prefix compaction,
//     no atomics, so downstream float sums are reproducible). Entry = {atom index in molecule, packed ranges}:
//     admitted y range in SUBY-voxel slabs (lo | hi << 8) and admitted z range in SUBZ-voxel sub-tiles
//     (lo << 16 | hi << 24) — all the later slab / sub-tile filters need (D <= 1024). List (b, sx) lives at
//     xlist[(a0 + 2*b) * nsx + sx * (N_b + 2)] (a0 = first atom, N_b = atoms of molecule b: regions are packed,
//     so ragged batches cost sum(N) entries per x-slab); entry 0 = {count, EMPTY}, entry 1 = {a0, EMPTY}.
//  B. slab lists: for every slab (sy, zc) of this x-slab the x-list is compacted once more against the slab's
//     y/z box: slist[slab * SLOTS] = {count, first atom}, then up to SLOTS-1 entries. The voxelize kernel reads
//     512 B of it per 63 candidates instead of scanning; a count above SLOTS-1 (LINE_OVERFLOW) sends that slab
//     to the x-list path.
constexpr int XL_HEADER = 2;
constexpr int XL_LDS = 1024; // x-list entries cached in LDS for pass B (8 KB; + 8 KB of lines: 8 blocks per CU); longer lists are re-read from L2
constexpr int SLOTS = 64;      // primary slab line: header + 63 candidates = 512 B, one per slab, densely packed
constexpr int EXT_SLOTS = 192; // extension line (entries 64..255) in a separate array: touched only by dense slabs
constexpr int LINE_CAP = SLOTS + EXT_SLOTS - 1; // candidates a slab can hold before it takes the x-list path
constexpr unsigned LINE_OVERFLOW = 0xffffffffu;
static_assert(SLOTS == SLAB_LINE_ENTRIES && EXT_SLOTS == SLAB_EXT_ENTRIES, "slab line sizes are shared with the host side");

// Configurations (launch_xbin): THREADS = 256 for batches (8 blocks per compute unit; CH = 4 / 8 / 16 chunks of 64 atoms
// per wave and round, so that molecules of up to 4 096 atoms need one round of pass A; NQ = 4 lines per wave and sweep);
// 64 (one wave per block) for batches of small molecules: the same passes with a quarter of the waves, which is what
// the 4 096 blocks of a ligand batch are bound by; 1 024 for one or a few large molecules, where the kernel is a chain
// of latencies on a mostly idle chip: every key of the molecule is requested at once (CH up to 16 x 1 024 atoms; with
// 256 threads and 4 chunks the 10 000 atoms of cfg-5 took ten rounds of one exposed memory latency each, 13 of the
// kernel's 20 us), the whole x-list stays in LDS (XLN), and each of the 16 waves builds NQ = 1 line per sweep (several
// blocks per x-slab) or, for small batches, NQ = 4 (one block per (molecule, x-slab)).
template <int THREADS, int XLN, int CH, int NQ>
__global__ void __launch_bounds__(THREADS)
    xbin_kernel(const uint2 *__restrict__ xp, const int64_t *__restrict__ offsets, int64_t n_one, int b0, int nsx, int nsy, int nzc, int NW,
                uint2 *__restrict__ xlist, uint2 *__restrict__ slist, uint2 *__restrict__ slist_ext) {
    __shared__ uint2 xs[XLN]; // (one-wave blocks serve molecules of <= 256 atoms: XLN = 256)
    constexpr int NWV = THREADS / 64; // waves per block
    __shared__ int wcnt[2][NWV];
    __shared__ int any_overflow;
    __shared__ uint2 line[NWV][NQ * SLOTS]; // the NQ slab lines each wave is building
    const int b = b0 + blockIdx.x / nsx, sx = blockIdx.x % nsx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t a0 = offsets ? offsets[b] : 0, a1 = offsets ? offsets[b + 1] : n_one; // (null: one molecule of n_one atoms)
    const int x0 = SUBX * sx;
    if (tid == 0) any_overflow = 0;
    uint2 *dst = xlist + ((size_t)a0 + 2 * (size_t)b) * nsx + (size_t)sx * (size_t)(a1 - a0 + XL_HEADER);
    int count = 0, phase = 0;
    XB_STAMP(0);
    if (a1 > a0) {
        // A round takes nch <= CH chunks of 64 consecutive atoms per wave (the last round only as many as are left):
        // wave w owns atoms [64*nch*w, 64*nch*(w+1)) of the round, so list order = atom order needs only one number per
        // wave from the others (its total); the chunk offsets are the wave's own popcounts.
        const int n = (int)(a1 - a0); // (a molecule's atoms are indexed in 32 bits)
        const uint2 *__restrict__ xpm = xp + a0;
        auto chunks_of = [&](const int rbase) {
            const int left = n - rbase;
            return left >= CH * THREADS ? CH : (left + THREADS - 1) / THREADS;
        };
        auto fetch = [&](uint2 (&v)[CH], const int rbase) {
            const int nch = chunks_of(rbase), first = rbase + wave * nch * 64 + lane;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int i = first + u * 64; // clamped (a select between addresses would make it a flat load); round() masks
                v[u] = xpm[i < n ? i : n - 1]; // (no branch per load: the compiler would wait for each one at its join)
            }
        };
        auto round = [&](const uint2 (&v)[CH], const int rbase) {
            const int nch = chunks_of(rbase), first = rbase + wave * nch * 64 + lane;
            // This wave's atoms of the round end at rend (chunks u >= nch would reach into the next wave's). The three
            // conditions - an atom of this wave, x range reaching the slab from below and from above - are differences
            // that must all be non-negative: one vector comparison of their OR gives the ballot mask directly (a
            // conjunction of three comparisons is three masks and two scalar ANDs, and this kernel is bound by the scalar
            // unit: 680 scalar against 480 vector instructions per wave at 256 molecules).
            const int wend = rbase + (wave + 1) * nch * 64, rend = wend < n ? wend : n;
            bool m[CH];
            int cnt[CH], own = 0;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int lo = (int)(v[u].x & 0xffff), hi = (int)(v[u].x >> 16);
                m[u] = (((x0 + SUBX - 1) - lo) | (hi - x0) | (rend - 1 - (first + u * 64))) >= 0;
                cnt[u] = __popcll(__ballot(m[u]));
                own += cnt[u];
            }
            if (lane == 0) wcnt[phase & 1][wave] = own;
            if (phase == 0) XB_STAMP(6); // wave 0: loads arrived, matches counted
            __syncthreads();
            int at = count;
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                const int c = wcnt[phase & 1][w];
                at += (w < wave) ? c : 0;
                count += c;
            }
            auto scatter = [&](auto lds_only) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (m[u]) {
                        const unsigned long long mk = __builtin_amdgcn_read_exec(); // == ballot(m[u]) in here
                        const int pos = at + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                        const uint2 en = make_uint2((unsigned)(first + u * 64), v[u].y);
                        if (decltype(lds_only)::value || pos < XLN) xs[pos] = en;
                        else dst[XL_HEADER + pos] = en; // beyond the LDS copy: straight to the global list
                    }
                    at += cnt[u];
                }
            };
            if (count <= XLN) scatter(std::true_type{}); // (block-uniform) the common case: the whole list fits the LDS copy
            else scatter(std::false_type{});
            ++phase;
        };
        // One register set and no prefetch across rounds: with the loads of two rounds in flight the compiler waits for
        // the older ones before it issues the newer (register reuse across the loop's back edge), and guards around
        // single loads end in a wait at every join. The launcher picks CH so that most molecules need one round.
        uint2 v[CH];
        for (int rbase = 0; rbase < n; rbase += CH * THREADS) {
            fetch(v, rbase);
            round(v, rbase);
        }
    }
    // the tail of a long x-list is read back by this block in pass B: workgroup-scope release here, workgroup-scope
    // loads there (an agent-scope fence makes every block write its XCD's L2 back: measured 6x on this kernel)
    __threadfence_block();
    __syncthreads();
    XB_STAMP(1); // pass A done

    const int nslab = nsy * nzc;
    const size_t xslab = (size_t)b * nsx + sx;
    uint2 *sl_base = slist + xslab * (size_t)nslab * SLOTS;
    uint2 *ext_base = slist_ext + xslab * (size_t)nslab * EXT_SLOTS;
    const int nlds = count < XLN ? count : XLN;
    // each wave builds NQ slab lines per pass over the x-list (one LDS read per round serves all of them)
    // (blockIdx.y splits the slabs of one x-slab over gridDim.y blocks when a grid has many slabs per x-slab)
    for (int g = NQ * wave + NQ * NWV * (int)blockIdx.y; g < nslab; g += NQ * NWV * (int)gridDim.y) {
        int sy[NQ], zt_lo[NQ], zt_hi[NQ], n[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int sl = g + q;
            sy[q] = (sl < nslab) ? sl / nzc : 255; // 255: beyond the grid, matches no entry
            zt_lo[q] = (sl - (sl / nzc) * nzc) * NW;
            zt_hi[q] = zt_lo[q] + NW - 1;
            n[q] = 0;
        }
        uint2 *ln = line[wave];
        // 64 x-list entries -> appended, in order, to the lines of the slabs they touch. ZT: the z test (grids of one slab per
        // row - nzc == 1 - need none: every listed atom's z range lies in the grid). EXT: the second sweep of a wave that
        // met a line with more than SLOTS-1 candidates; it writes the entries 63.. to the slab's extension line, so that
        // the common sweep carries no code for them.
        auto take = [&](const uint2 en, auto zt, auto ext_pass) {
            constexpr bool ZT = decltype(zt)::value, EXT = decltype(ext_pass)::value;
            const unsigned pk = en.y;
            const int ylo = (int)(pk & 0xff), yhi = (int)((pk >> 8) & 0xff), zlo = (int)((pk >> 16) & 0xff), zhi = (int)(pk >> 24);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                // (all of: slab's y inside [ylo, yhi], z ranges overlap - as one sign test, see pass A)
                int t = (sy[q] - ylo) | (yhi - sy[q]);
                if (ZT) t |= (zt_hi[q] - zlo) | (zhi - zt_lo[q]);
                const bool mm = t >= 0;
                const unsigned long long mk = __ballot(mm);
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, (unsigned)n[q]));
                if (!EXT) {
                    if (mm & (pos < SLOTS - 1)) ln[q * SLOTS + 1 + pos] = en;
                } else {
                    if (mm & (pos >= SLOTS - 1) & (pos < LINE_CAP)) ext_base[(size_t)(g + q) * EXT_SLOTS + (pos - (SLOTS - 1))] = en;
                }
                n[q] += __popcll(mk);
            }
        };
        auto sweep = [&](auto zt, auto ext_pass) {
            // (two loops: a global load inside the common LDS loop would put a vmcnt(0) wait, i.e. a wait for the
            // previous round's stores, into every round)
#pragma unroll 2
            for (int i0 = 0; i0 < nlds; i0 += 64) {
                const int i = i0 + lane;
                take(i < nlds ? xs[i] : make_uint2(0u, EMPTY_ENTRY), zt, ext_pass);
            }
            for (int i0 = XLN; i0 < count; i0 += 64) { // beyond the LDS copy: this block's own stores, read back
                const int i = i0 + lane;
                uint2 en = make_uint2(0u, EMPTY_ENTRY);
                if (i < count) {
                    en.x = __hip_atomic_load(&dst[XL_HEADER + i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    en.y = __hip_atomic_load(&dst[XL_HEADER + i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                take(en, zt, ext_pass);
            }
        };
        if (nzc > 1) sweep(std::true_type{}, std::false_type{});
        else sweep(std::false_type{}, std::false_type{});
        XB_STAMP(2); // pass B (last group of wave 0)
        bool full = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) full = full || n[q] > SLOTS - 1;
        if (full) { // (wave-uniform, rare)
#pragma unroll
            for (int q = 0; q < NQ; ++q) n[q] = 0;
            sweep(std::true_type{}, std::true_type{});
        }
        XB_STAMP(3); // ... and the read-back tail
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (g + q >= nslab) break;
            // more candidates than the line and its extension hold: the slab takes the x-list path
            const unsigned hdr = (n[q] > LINE_CAP) ? LINE_OVERFLOW : (unsigned)n[q];
            if (lane == 0) {
                ln[q * SLOTS] = make_uint2(hdr, (unsigned)a0);
                if (n[q] > LINE_CAP) {
                    any_overflow = 1;
                    // an overflowing slab's line carries, in entry 1, where its (molecule, x-slab) list lives: the voxelize
                    // kernel then needs no list base, no offsets and no atom count for this rare path (kernel arguments
                    // it would hold in scalar registers through its whole hot path)
                    const unsigned long long at = (unsigned long long)reinterpret_cast<uintptr_t>(dst);
                    ln[q * SLOTS + 1] = make_uint2((unsigned)at, (unsigned)(at >> 32));
                }
            }
            // the primary line leaves as one store of the used part, rounded up to 32 B (entries past the count are
            // never interpreted): an empty slab costs 32 B, not 512. (nt stores: the line then misses L2 in the
            // voxelize kernel, slower overall.)
            if (lane < ((n[q] + 4) & ~3)) sl_base[(size_t)(g + q) * SLOTS + lane] = ln[q * SLOTS + lane];
        }
    }
    // the global x-list is only read by slabs on the x-list path: publish the LDS part when one exists
    XB_STAMP(4); // wave 0's lines stored
    __syncthreads();
    XB_STAMP(5);
#ifdef MVX_DIAG
    if (g_diag_xb && tid == 0) g_diag_xb[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 7] = (unsigned long long)count;
#endif
    if (any_overflow) {
        if (tid == 0) dst[0] = make_uint2((unsigned)count, EMPTY_ENTRY);
        if (tid == 1) dst[1] = make_uint2((unsigned)a0, EMPTY_ENTRY);
        const int nl = count < XLN ? count : XLN;
        for (int i = tid; i < nl; i += THREADS) dst[XL_HEADER + i] = xs[i];
    }
}

hipError_t launch_xbin(const uint2 *xp, const int64_t *offsets, int64_t n_one, int32_t b0, int32_t nb, int64_t max_atoms, int32_t nsx, int32_t nsy,
                       int32_t nzc, int32_t NW, uint2 *xlist, uint2 *slist, uint2 *slist_ext, hipStream_t s) {
    if (nb <= 0) return hipSuccess;
    const int nslab = nsy * nzc;
    if (max_atoms <= 256) { // small molecules (one round of pass A for a single wave): one-wave blocks
        int parts = 1;
        while (parts * 4 < nslab && parts < 4 && (long long)nb * nsx * parts < 8192) parts *= 2;
        hipLaunchKernelGGL((xbin_kernel<64, 256, 4, 4>), dim3((unsigned)(nb * nsx), (unsigned)parts), dim3(64), 0, s, xp, offsets, n_one, b0, nsx,
                           nsy, nzc, NW, xlist, slist, slist_ext);
        return hipGetLastError();
    }
    // one block builds 16 slab lines per pass; grids with more slabs per x-slab (D > 64) and few molecules get
    // several blocks per (molecule, x-slab), each repeating the cheap pass A, until ~2048 blocks are in flight
    int parts = 1;
    while (parts * 16 < nslab && (long long)nb * nsx * parts < 2048) parts *= 2;
    // Large molecules, few of them (at most two 1024-thread blocks per compute unit): latency is all there is. One
    // block per (molecule, x-slab) with four lines per wave when that already gives >= 256 blocks (pass A runs once per
    // pair), else one line per wave and nslab/16 blocks per pair (each repeats pass A on an otherwise idle unit).
    if (max_atoms > 2048) {
        const long long pairs = (long long)nb * nsx;
        const int parts1 = (nslab + 15) / 16, parts4 = (nslab + 63) / 64;
        const bool four = pairs * parts1 > 256;
        const int bparts = four ? parts4 : parts1;
        if (pairs * bparts <= 512) {
            const dim3 grid((unsigned)pairs, (unsigned)bparts);
#define MVX_XBIN_BIG(CHUNKS)                                                                                                          \
    do {                                                                                                                              \
        if (four)                                                                                                                     \
            hipLaunchKernelGGL((xbin_kernel<1024, 2 * XL_LDS, CHUNKS, 4>), grid, dim3(1024), 0, s, xp, offsets, n_one, b0, nsx, nsy, nzc, \
                               NW, xlist, slist, slist_ext);                                                                          \
        else                                                                                                                          \
            hipLaunchKernelGGL((xbin_kernel<1024, 4 * XL_LDS, CHUNKS, 1>), grid, dim3(1024), 0, s, xp, offsets, n_one, b0, nsx, nsy, nzc, \
                               NW, xlist, slist, slist_ext);                                                                          \
    } while (0)
            if (max_atoms <= 4 * 1024) MVX_XBIN_BIG(4); // all of the largest molecule's atoms in flight at once when <= 16 384
            else if (max_atoms <= 6 * 1024) MVX_XBIN_BIG(6);
            else if (max_atoms <= 8 * 1024) MVX_XBIN_BIG(8);
            else if (max_atoms <= 10 * 1024) MVX_XBIN_BIG(10);
            else if (max_atoms <= 12 * 1024) MVX_XBIN_BIG(12);
            else MVX_XBIN_BIG(16);
#undef MVX_XBIN_BIG
            return hipGetLastError();
        }
    }
    // chunks per round: one round of pass A (one barrier, one exchange of counts) for molecules of up to 4 096 atoms
    // (cfg-2, 256 molecules: 56.4 us with 4 chunks = four rounds, 50.7 with 8, 48.6 with 16)
    const dim3 grid((unsigned)(nb * nsx), (unsigned)parts);
#define MVX_XBIN_256(CHUNKS)                                                                                                          \
    hipLaunchKernelGGL((xbin_kernel<256, XL_LDS, CHUNKS, 4>), grid, dim3(256), 0, s, xp, offsets, n_one, b0, nsx, nsy, nzc, NW, xlist, \
                       slist, slist_ext)
    if (max_atoms <= 1024) MVX_XBIN_256(4);
    else if (max_atoms <= 2048) MVX_XBIN_256(8);
    else MVX_XBIN_256(16);
#undef MVX_XBIN_256
    return hipGetLastError();
}

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

#ifndef MVX_EMPTY_SPLIT // write_slab: units of 64 cycles between the pieces of an empty slab's zero fill (paced launches)
#define MVX_EMPTY_SPLIT 20
#endif
#ifndef MVX_RSLEEP // OpsMx32::write: units of 64 cycles a wave waits after each write-out round of a paced slab
#define MVX_RSLEEP 20
#endif

// 16-B output store: non-temporal. Output bytes are written once and never re-read here; nt keeps them from displacing
// the re-read inputs in L2 (0.69 -> 0.54 ms, cfg-2; sc1 = plain). A/B builds (make EXPERIMENT=1) pick the kind at run
// time - 0: plain (line stays in the XCD's L2); 1: nt; 2: sc1 (write-through); the shipped kernels hold one store form.
__device__ __forceinline__ void store_f4(float *dst, const float4 v, int kind) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x, v.y, v.z, v.w};
#ifdef MVX_EXPERIMENT
    if (kind == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(x) : "memory");
        return;
    }
    if (kind == 0) {
        *reinterpret_cast<f4 *>(dst) = x;
        return;
    }
#else
    (void)kind;
#endif
    __builtin_nontemporal_store(x, reinterpret_cast<f4 *>(dst));
}

// floats per tile row: SUBZ*NW plus a pad that keeps ds_write_b32 conflict-free for the lane -> (row, column) map
__host__ __device__ __forceinline__ int row_stride_floats(int NW) { return SUBZ * NW + 8; }
// words per staged row: 16 of record + the channel weights, padded to an ODD number of 16-B quads - the row filter reads one
// row per lane (ds_read_b128 at a lane stride of one row): with 12 quads per row (CT = 32) sixteen lanes fell on four
// distinct bank groups (4-way conflict, and 16-way for the 4-byte read of the z range); with 13 they are conflict-free
__host__ __device__ constexpr int cand_stride_words(int ct) {
    const int w = 16 + (ct < 4 ? 4 : ct);
#ifdef MVX_SW_PLAIN // (A/B builds)
    return w;
#endif
    return ((w / 4) & 1) ? w : w + 4;
}

#ifndef MVX_CR
#define MVX_CR 4 // channels per transposition round of the float32 write-out: 512 threads read back exactly one 32-row tile
                 // (cfg-2 x 256, same box: 16 -> 0.783-0.787 of peak, 8 -> 0.790, 4 -> 0.792-0.795: smaller store bursts interleave better)
#endif
constexpr int DIRECT_CR = 16; // write-out rounds of voxelize_direct_kernel
size_t voxelize_lds_bytes(int32_t ct, int32_t NW, int32_t crmax) {
    const int cr = ct < crmax ? ct : crmax;
    const size_t tile = (size_t)cr * RPC * row_stride_floats(NW) * 4;
    const size_t cand = (size_t)64 * cand_stride_words(ct) * 4;
    return tile > cand ? tile : cand;
}
// voxelize_kernel's row / tile region (-DMVX_PREFETCH: two row buffers, the second one receives the next round's rows by LDS-DMA)
size_t voxelize_rounds_lds_bytes(int32_t ct, int32_t NW, int32_t crmax) {
    const size_t one = voxelize_lds_bytes(ct, NW, crmax);
#ifdef MVX_PREFETCH
    const size_t rows2 = (size_t)2 * 64 * cand_stride_words(ct) * 4;
    return one > rows2 ? one : rows2;
#else
    return one;
#endif
}

// dense kernel: candidate rows staged per round = what fits in the out tile's bytes, at least 64, at most LCAP
int32_t voxelize_dcap(int32_t ct, int32_t NW) {
    const int cr = ct < MVX_CR ? ct : MVX_CR;
    const int lcap = 64 * (NW < 4 ? NW : 4);
    const size_t tile = (size_t)cr * RPC * row_stride_floats(NW) * 4;
    int cap = (int)(tile / ((size_t)cand_stride_words(ct) * 4));
    if (cap < 64) cap = 64;
    if (cap > lcap && lcap >= 64) cap = lcap;
    return cap;
}

size_t dense_lds_bytes(int32_t ct, int32_t NW) {
    const int cr = ct < MVX_CR ? ct : MVX_CR;
    const int lcap = 64 * (NW < 4 ? NW : 4);
    const size_t tile = (size_t)cr * RPC * row_stride_floats(NW) * 4;
    const size_t cand = (size_t)voxelize_dcap(ct, NW) * cand_stride_words(ct) * 4;
    return (size_t)8 * lcap + 16 + (tile > cand ? tile : cand);
}

constexpr int CR64 = 8; // float64 write-out: channels per transposition round
__host__ __device__ __forceinline__ int row_stride_doubles(int NW) { return SUBZ * NW + 8; }
// float64 kernel: rows of 16 + 2*ct words, 64 per round (p.dcap = 64), no out tile
size_t dense64_lds_bytes(int32_t ct, int32_t NW) {
    const int lcap = 64 * (NW < 4 ? NW : 4);
    const size_t rows = (size_t)64 * (16 + 2 * ct) * 4;
    const size_t tile = (size_t)(ct < CR64 ? ct : CR64) * RPC * row_stride_doubles(NW) * 8;
    return (size_t)8 * lcap + 16 + (rows > tile ? rows : tile);
}

// what a lane knows about its voxel and its workgroup's slab
struct LaneCtx {
    double gx, gy, gz; // voxel centre: axis[i] = i*res - width/2 (numpy/voxelizer.py:41-43)
    double gx1;        // OpsMx32 / OpsMx64 (several voxels per lane): the x + 1 plane's coordinate
    double gy1;        // OpsMx64 only (four voxels per lane: x, x + 1 times iy, iy + 2): the second y row's coordinate
    int grp;           // grouped launches (channel-wise features by radius) only: the radius slot of channel cbase + lane % 32
                       // (-1: no such channel), the slots present in this chunk (bit mask, uniform) and the LDS copy of
    unsigned gmask;    // the slots' {T, k}
    const double *gtab;
    int ix, iy, iz;
    int zt_w;          // this wave's sub-tile index along z
    int cbase;         // first channel of this workgroup's chunk
};

// One candidate (row r staged in LDS) into the accumulators.
template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
__device__ __forceinline__ void accumulate_row(float2v (&acc)[(CT + 1) / 2], const unsigned *r, const LaneCtx &L, int C,
                                               const double *__restrict__ Tc, const float *__restrict__ kc) {
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
    float val = 0.0f;
    if (!CHANWISE) {
        // no early-out on "no lane hit": ~85 % of the filtered candidates hit, and a straight-line body pipelines
        const float ev = GAUSS ? __builtin_amdgcn_exp2f(__uint_as_float(q.x) * d2f) : 1.0f;
        val = hit ? ev : 0.0f;
    }
    if constexpr (CHANWISE) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int ch = (L.cbase + c < C) ? L.cbase + c : C - 1;
            const float ev = GAUSS ? __builtin_amdgcn_exp2f(kc[ch] * d2f) : 1.0f;
            const float vc = (hit && d2 <= Tc[ch]) ? ev : 0.0f;
            if (c & 1) acc[c / 2].y = fmaf(vc, f[c], acc[c / 2].y);
            else acc[c / 2].x = fmaf(vc, f[c], acc[c / 2].x);
        }
    } else if constexpr (CT == 1) {
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
// RUNS: the kernel also serves grids whose rows are not whole 16-byte quads (store_runs). Only the per-lane-range kernels
// are compiled with it (the host sends such grids there): the aligned-grid kernels keep their register budget.
template <int CT, bool RUNS, int CRMAX = MVX_CR>
__device__ __forceinline__ void write_slab(const float2v (&acc)[(CT + 1) / 2], bool any, float *tile, int tid, int lane,
                                           int wave, int NW, int b, int cbase, int x0, int y0, int z0, float *out,
                                           const VoxParams &P) {
    constexpr int CR = CT < CRMAX ? CT : CRMAX; // channels per write-out round
    constexpr int NROUND = CT / CR;
    const int D = P.D;
    const int RS = row_stride_floats(NW);
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    const int F4 = (SUBZ / 4) * NW; // float4 slots per row
    const int q = tid % F4;         // float4 slot inside a row
    const int rfirst = tid / F4;    // row of this thread in pass 0; rows advance by 4 channels (4*RPC rows) per pass
    const int zq = z0 + 4 * q;
    const int sxx = (rfirst >> SUBY_SH) & (SUBX - 1), syy = rfirst & (SUBY - 1), cfirst = rfirst / RPC;
    const bool vox_ok = (x0 + sxx < D) && (y0 + syy < D) && (zq < D);
    float *dst0 = out + ((size_t)b * P.C + cbase + cfirst) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
    if (!any) {
        // Pacing: a workgroup that has nothing to compute would fire its 64 KB of stores the moment it starts; holding
        // them back ~1.7 us (4096 cycles) lets the store streams of the resident workgroups interleave: ligand batches
        // 6.2 -> 6.7 TB/s (sleep 16 / 32 / 48 / 64 / 80 / 100: +0.9 / 2.8 / 4.7 / 7.5 / 7.0 / 3.7 %).
        // (only when several rounds of workgroups follow each other; a small launch would just start later)
        if (P.pace) __builtin_amdgcn_s_sleep(64);
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
                if (c < CT && cbase + c < P.C) store_f4(dst0 + (size_t)(4 * p) * D3, make_float4(0.f, 0.f, 0.f, 0.f), P.store_kind);
                // ... and the fill itself goes out in pieces of two store instructions (16 KB per workgroup) ~1300 cycles
                // apart, like the write-out rounds of OpsMx32::write: ligand batches 6.55 -> 6.83 TB/s (0.85 of peak;
                // 512 / 1024 / 1536 / 2048 cycles: +2 / +3.5 / +4.3 / +3.6 %; with a first wait of 2048 instead of 4096
                // cycles: -1 / +1 %)
                if (P.pace && (p & 1) && p + 1 < (CT + 3) / 4) __builtin_amdgcn_s_sleep(MVX_EMPTY_SPLIT);
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
                    store_f4(dst0 + (size_t)(rd * CR + 4 * p) * D3, v, P.store_kind);
                }
            }
        }
    }
}

// float64 write-out: the same transposition through an LDS tile, 8 channels per round, rows of SUBZ*NW doubles read
// back as 16-B pairs: 32 lanes per 512-B row instead of eight 64-B runs per store instruction straight from registers
// (2.5 -> TB/s on cfg-2). Begins with a barrier (the region may still hold candidate rows) and ends without one.
template <int CT>
__device__ __forceinline__ void write_slab64(const double (&acc)[CT], double *tile, int tid, int lane, int wave, int NW, int b,
                                             int cbase, int x0, int y0, int z0, double *out, const VoxParams &P) {
    constexpr int CR = CT < CR64 ? CT : CR64;
    constexpr int NROUND = (CT + CR - 1) / CR;
    const int D = P.D;
    const int RS = row_stride_doubles(NW);
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    const int F2 = (SUBZ / 2) * NW;   // 16-B slots per row
    const int nthr = NW * 64;
    const int rows_per_pass = nthr / F2; // 16 for any NW
    const int q = tid % F2, rfirst = tid / F2;
    const int zq = z0 + 2 * q;
    const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1), lx = lane >> (SUBZ_SH + SUBY_SH);
    const int col = SUBZ * wave + lz;
    const int rxy = lx * SUBY + ly;
    const bool vec = P.vec_store != 0; // D even and a 16-B aligned grid
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        __syncthreads(); // candidate rows (round 0) / previous tile (later rounds) fully consumed
#pragma unroll
        for (int c = 0; c < CR; ++c)
            if (rd * CR + c < CT) tile[(c * RPC + rxy) * RS + col] = acc[rd * CR + c];
        __syncthreads();
        for (int row = rfirst; row < CR * RPC; row += rows_per_pass) {
            const int c = row / RPC, r = row - c * RPC;
            const int sxx = (r >> SUBY_SH) & (SUBX - 1), syy = r & (SUBY - 1);
            const int ch = cbase + rd * CR + c;
            if (rd * CR + c < CT && ch < P.C && x0 + sxx < D && y0 + syy < D && zq < D) {
                const double2 v = *reinterpret_cast<const double2 *>(tile + row * RS + 2 * q);
                double *dst = out + ((size_t)b * P.C + ch) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
                if (vec) {
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store((d2v){v.x, v.y}, reinterpret_cast<d2v *>(dst));
                } else {
                    dst[0] = v.x;
                    if (zq + 1 < D) dst[1] = v.y;
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

// float64 grids (precision = 64): the reference then keeps distances, ratios, densities and sums in float64
// (numpy/voxelizer.py:33-34, 544-560). Membership sqrt_f64(d2) / r <= 1 is decided exactly by d2 <= T (d2_threshold64,
// T in the record's T slot; per channel for channel-wise radii), so misses cost no sqrt / division; the gaussian value
// is exp(c * d2) (gauss_coeff64, c in the record's last two words), evaluated only when some lane of the wave hits.
template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
__device__ __forceinline__ void accumulate_row64(double (&acc)[CT], const unsigned *r, const LaneCtx &L, int C,
                                                 const double *__restrict__ Tc, const double *__restrict__ kc) {
    const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
    const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
    const double dx = Pxy.x - L.gx, dy = Pxy.y - L.gy, dz = PzT.x - L.gz;
    const double d2 = (dx * dx + dy * dy) + dz * dz; // cdist order, no fma
    bool in_range = true;
    if (LANE_RANGE) {
        const uint4 q = *reinterpret_cast<const uint4 *>(r + 8);
        const unsigned zr = r[12];
        in_range = (L.ix >= (int)(q.z & 0xffff)) && (L.ix <= (int)(q.z >> 16)) && (L.iy >= (int)(q.w & 0xffff)) &&
                   (L.iy <= (int)(q.w >> 16)) && (L.iz >= (int)(zr & 0xffff)) && (L.iz <= (int)(zr >> 16));
    }
    const double *f = reinterpret_cast<const double *>(r + 16);
    if constexpr (CHANWISE) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int ch = (L.cbase + c < C) ? L.cbase + c : C - 1;
            const bool hit = in_range && d2 <= Tc[ch];
            double val = 0.0;
            if (hit) val = GAUSS ? exp(kc[ch] * d2) : 1.0;
            acc[c] = fma(val, f[c], acc[c]);
        }
    } else {
        const bool hit = in_range && d2 <= PzT.y;
        double val = 0.0;
        if (hit) val = GAUSS ? exp(*reinterpret_cast<const double *>(r + 14) * d2) : 1.0;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = fma(val, f[c], acc[c]);
    }
}

// What differs between float32 and float64 grids: accumulator type, staged row width, the per-candidate update
// and the write-out. OpsF32 is the tuned path; OpsF64 favours exactness over speed (8-B stores straight from
// registers, no LDS transposition).
template <int CT_, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
struct OpsF32 {
    static constexpr bool RUNS = LANE_RANGE; // carries the run-wise write-out (store_runs)
    static constexpr int CT = CT_;
    static constexpr bool GROUPED = false;
    typedef float2v Acc[(CT + 1) / 2];
    static constexpr int WORDS = 1;                   // 32-bit words per channel weight
    static constexpr int WW = CT;                     // weight words staged per row
    static constexpr int SW = cand_stride_words(CT);  // row stride in LDS, words
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int c = 0; c < (CT + 1) / 2; ++c) acc[c] = (float2v){0.0f, 0.0f};
    }
    static __device__ __forceinline__ void accumulate(Acc &acc, const unsigned *r, const LaneCtx &L, const VoxParams &P,
                                                      const double *__restrict__ Tc, const float *__restrict__ kc) {
        accumulate_row<CT, GAUSS, CHANWISE, LANE_RANGE>(acc, r, L, P.C, Tc, kc);
    }
    static __device__ __forceinline__ LaneCtx ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase, const VoxParams &P) {
        return make_lane_ctx(lane, wave, x0, y0, z0, zt_lo, cbase, P);
    }
    // the rows of `mask` (one bit per staged row, atom order) into the accumulators
    static __device__ __forceinline__ void walk(Acc &acc, unsigned long long mask, const unsigned *un, int lane, const LaneCtx &L,
                                                const VoxParams &P, const double *__restrict__ Tc, const float *__restrict__ kc) {
        while (mask) {
            const int sl = __builtin_ctzll(mask);
            mask &= mask - 1;
            accumulate(acc, un + sl * SW, L, P, Tc, kc);
        }
    }
    static __device__ __forceinline__ void write(const Acc &acc, bool any, unsigned *un, int tid, int lane, int wave, int NW,
                                                 int b, const LaneCtx &L, int x0, int y0, int z0, void *out, const VoxParams &P) {
        write_slab<CT, LANE_RANGE>(acc, any, reinterpret_cast<float *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0,
                       static_cast<float *>(out), P);
    }
    // per-molecule launches (voxelize_direct_kernel): 16 channels per round - two rounds, four barriers; the small
    // rounds pay when thousands of workgroups' store bursts interleave, not when 512 workgroups store once (cfg-2
    // single call 21.2 -> 20.4 us)
    static __device__ __forceinline__ void write_wide(const Acc &acc, bool any, unsigned *un, int tid, int lane, int wave, int NW,
                                                      int b, const LaneCtx &L, int x0, int y0, int z0, void *out, const VoxParams &P) {
        write_slab<CT, LANE_RANGE, DIRECT_CR>(acc, any, reinterpret_cast<float *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0,
                                  static_cast<float *>(out), P);
    }
};

// ---- 32 channels on the matrix cores ---------------------------------------------------------------------------------
// What a wave does per candidate is a rank-1 update of its (32 channels x 64 voxels) tile: acc[c][v] += w[c] * val[v] -
// the reference's own formulation is a matmul (numpy/voxelizer.py:232-235). On the vector ALU that costs, per candidate
// and wave, 16 v_pk_fma_f32 (64 issue cycles) and eight 16-B LDS broadcast reads of the weight row (32 LDS cycles: a
// ds_read_b128 takes 4 cycles whether or not its 64 lanes read the same address), and these two are what bounds the
// kernel once slabs hold more than ~60 candidates (radii >= 1.5 A on a 0.5 A grid; rocprofv3 counters in
// profiles/r03_radius_pmc.txt: LDS array 66 % and vector ALU 62 % busy at 2.0 A, 0.41 of the HBM peak).
// v_mfma_f32_32x32x2_f32 does the same update for TWO candidates in float32 - D = fma(a1, b1, fma(a0, b0, C)), one
// rounding per step, k = 0 first (probed on the hardware: tools/micro/mfma_layout.hip), i.e. bit for bit the chain of
// fmaf in candidate order that the vector path evaluates - on the matrix pipe, which runs beside the vector ALU, and it
// takes its operands one dword per lane: A[i = lane % 32][k = lane / 32], B[k = lane / 32][j = lane % 32]. With
// A = weights (i = channel) and B = values (j = voxel):
//   * lane l evaluates candidate k = l / 32 of the pair for TWO voxels, (x0, ly, lz) and (x0 + 1, ly, lz) with
//     (ly, lz) = ((l % 32) / 8, l % 8): the same fp64 d2 / threshold / exp2 work per (voxel, candidate) as before
//     (dy^2 and dz^2 are shared by the two voxels), two MFMAs per pair (x plane and x + 1 plane);
//   * the weight operand is ONE 4-byte LDS read per pair (lane l: weight l % 32 of its candidate's row) instead of
//     sixteen 16-byte broadcast reads: LDS cycles per candidate 44 -> 6.
// D[i][j] comes out with lane l holding voxel j = l % 32 - the (ly, lz) it evaluated - for the channels
// i = (r % 4) + 8 (r / 4) + 4 (l / 32), r = 0..15: one voxel per lane as in the vector path, half the channels in each half
// of the wave. The write-out therefore keeps the vector path's tile ([channel][x, y row][z], eight channels per round
// here: channels 8m .. 8m+3 sit in registers 4m .. 4m+3 of lanes 0-31, channels 8m+4 .. 8m+7 in the same registers of
// lanes 32-63, so every lane writes 4 channels x 2 planes per round) and its read-back / store code unchanged.
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int MX_CR = 8; // channels per write-out round of the matrix-core path
size_t voxelize_mx_lds_bytes(int32_t NW) { return voxelize_rounds_lds_bytes(32, NW, MX_CR); }

template <bool GAUSS, bool LANE_RANGE, bool GROUPED_ = false, bool RUNS_ = LANE_RANGE>
struct OpsMx32 {
    static constexpr int CT = 32;
    static constexpr bool RUNS = RUNS_; // carries the run-wise write-out (store_runs)
    static constexpr bool GROUPED = GROUPED_;
    struct Acc {
        f16v p0, p1; // the x0 plane and the x0 + 1 plane of the sub-tile
    };
    static constexpr int WORDS = 1;
    static constexpr int WW = 32;
    static constexpr int SW = cand_stride_words(32);
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.p0[r] = acc.p1[r] = 0.0f;
    }
    static __device__ __forceinline__ LaneCtx ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase, const VoxParams &P) {
        const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1);
        LaneCtx L;
        L.ix = x0; // (x0 is a multiple of SUBX)
        L.iy = y0 + ly;
        L.iz = z0 + SUBZ * wave + lz;
        L.gx = (double)L.ix * P.res - P.half;
        L.gx1 = (double)(L.ix + 1) * P.res - P.half;
        L.gy = (double)L.iy * P.res - P.half;
        L.gz = (double)L.iz * P.res - P.half;
        L.zt_w = zt_lo + wave;
        L.cbase = cbase;
        return L;
    }
    static __device__ __forceinline__ void walk(Acc &acc, unsigned long long mask, const unsigned *un, int lane, const LaneCtx &L,
                                                const VoxParams &P, const double *__restrict__, const float *__restrict__) {
        const bool upper = lane >= 32; // this lane evaluates the pair's second candidate
        const int j = lane & 31;       // ... and feeds the weight of channel j of that candidate's row
        while (mask) {
            const int s0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            const bool two = mask != 0; // (uniform)
            int s1 = s0;
            if (two) {
                s1 = __builtin_ctzll(mask);
                mask &= mask - 1;
            }
            const bool valid = !upper || two; // an odd row count: the last pair's second half adds fma(0, 0, acc) = acc
            const unsigned *r = un + (upper ? s1 : s0) * SW;
            const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
            const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
            const double dx0 = Pxy.x - L.gx, dx1 = Pxy.x - L.gx1, dy = Pxy.y - L.gy, dz = PzT.x - L.gz;
            const double dy2 = dy * dy, dz2 = dz * dz;
            const double d2a = (dx0 * dx0 + dy2) + dz2; // cdist order, no fma
            const double d2b = (dx1 * dx1 + dy2) + dz2;
            bool hita = valid && d2a <= PzT.y, hitb = valid && d2b <= PzT.y;
            float k;
            if (LANE_RANGE) {
                const uint4 q = *reinterpret_cast<const uint4 *>(r + 8); // k, type, xr, yr
                const unsigned zr = r[12];
                k = __uint_as_float(q.x);
                const bool yz = (L.iy >= (int)(q.w & 0xffff)) && (L.iy <= (int)(q.w >> 16)) && (L.iz >= (int)(zr & 0xffff)) &&
                                (L.iz <= (int)(zr >> 16));
                hita = hita && yz && (L.ix >= (int)(q.z & 0xffff)) && (L.ix <= (int)(q.z >> 16));
                hitb = hitb && yz && (L.ix + 1 >= (int)(q.z & 0xffff)) && (L.ix + 1 <= (int)(q.z >> 16));
            } else {
                k = __uint_as_float(r[8]);
            }
            const float wj = valid ? __uint_as_float(r[16 + j]) : 0.0f;
            if constexpr (GROUPED) {
                // channel-wise radii: the record's radius is max(radii) (the culls' radius, numpy/voxelizer.py:138), so
                // hita / hitb so far only say "inside the largest ball" (and the index ranges). Per radius slot present
                // in this chunk: its own threshold and density for the same d2, the weight row masked to its channels -
                // channels of other slots receive fma(0, val, acc) = acc.
                const float d2fa = (float)d2a, d2fb = (float)d2b;
                unsigned todo = L.gmask;
                while (todo) {
                    const int g = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const double Tg = L.gtab[2 * g];
                    const float kg = reinterpret_cast<const float *>(L.gtab + 2 * g + 1)[0];
                    const float eva = GAUSS ? __builtin_amdgcn_exp2f(kg * d2fa) : 1.0f;
                    const float evb = GAUSS ? __builtin_amdgcn_exp2f(kg * d2fb) : 1.0f;
                    const float va = (hita && d2a <= Tg) ? eva : 0.0f, vb = (hitb && d2b <= Tg) ? evb : 0.0f;
                    const float wg = L.grp == g ? wj : 0.0f;
                    acc.p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wg, va, acc.p0, 0, 0, 0);
                    acc.p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wg, vb, acc.p1, 0, 0, 0);
                }
            } else {
                const float eva = GAUSS ? __builtin_amdgcn_exp2f(k * (float)d2a) : 1.0f;
                const float evb = GAUSS ? __builtin_amdgcn_exp2f(k * (float)d2b) : 1.0f;
                const float va = hita ? eva : 0.0f, vb = hitb ? evb : 0.0f;
                acc.p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wj, va, acc.p0, 0, 0, 0);
                acc.p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wj, vb, acc.p1, 0, 0, 0);
            }
        }
    }
    static __device__ __forceinline__ void write(const Acc &acc, int any, unsigned *un, int tid, int lane, int wave, int NW,
                                                 int b, const LaneCtx &L, int x0, int y0, int z0, void *out_, const VoxParams &P) {
        float *out = static_cast<float *>(out_);
        if (!any) { // zero fill without the LDS round trip: the one-voxel-per-lane code (no accumulator is read)
            float2v zero[16];
            write_slab<32, RUNS>(zero, false, reinterpret_cast<float *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0, out, P);
            return;
        }
        float *tile = reinterpret_cast<float *>(un);
        constexpr int CR = MX_CR, NROUND = 32 / CR;
        const int D = P.D;
        const int RS = row_stride_floats(NW);
        const size_t D2 = (size_t)D * D, D3 = D2 * D;
        // read-back exactly as write_slab: thread t takes float4 slot q of row rfirst (+ 4 channels per pass)
        const int F4 = (SUBZ / 4) * NW;
        const int q = tid % F4, rfirst = tid / F4, zq = z0 + 4 * q;
        const int sxx = (rfirst >> SUBY_SH) & (SUBX - 1), syy = rfirst & (SUBY - 1), cfirst = rfirst / RPC;
        const bool vox_ok = (x0 + sxx < D) && (y0 + syy < D) && (zq < D);
        float *dst0 = out + ((size_t)b * P.C + L.cbase + cfirst) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
        // this lane's voxel column in the tile, and the first of its four channels of a round
        const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1), h = lane >> 5;
        float *mine = tile + (4 * h * RPC + ly) * RS + SUBZ * wave + lz; // + (c * RPC + x * SUBY) * RS
        auto round = [&](auto rd_) {
            constexpr int rd = decltype(rd_)::value;
            __syncthreads(); // candidate rows (first round) / previous tile (later rounds) fully consumed
            if (rd == 0) VK_STAMP(4); // every wave's walk is done
            if (rd == 1) VK_STAMP(5); // round 0 transposed and its stores issued
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mine[(c * RPC) * RS] = acc.p0[4 * rd + c];
                mine[(c * RPC + SUBY) * RS] = acc.p1[4 * rd + c];
            }
            __syncthreads();
            if (vox_ok) {
#pragma unroll
                for (int p = 0; p < (CR + 3) / 4; ++p) {
                    const int c = cfirst + 4 * p; // channel inside the round
                    if (c < CR && L.cbase + rd * CR + c < P.C) {
                        const float4 v = *reinterpret_cast<const float4 *>(tile + (rfirst + 4 * RPC * p) * RS + 4 * q);
                        store_f4(dst0 + (size_t)(rd * CR + 4 * p) * D3, v, P.store_kind);
                    }
                }
            }
            // Pacing (any == 2: a slab of few candidates in a launch of at least 49 152 workgroups - the store-bound regime;
            // smaller launches lose 2-3 % by it, mvx_capi.hip): the
            // wave holds back ~1300 cycles after each round's stores, about the time a compute unit needs to drain the 16 KB
            // a workgroup just queued. Left alone a workgroup pushes its 64 KB within ~2 kcycles and the row loads of the
            // unit's other workgroups wait behind them. Same box, kernel, of peak: cfg-2 x 256 0.771-0.794 -> 0.807-0.811,
            // cfg-5 x 8 0.697 -> 0.711; 256 / 1024 / 1536 / 1920 / 2560 cycles: +0.5 / +2.3 / +2.5 / +0.5 / -7 %. Slabs
            // of more than MVX_PACE_MAX candidates are bound by their walk and lose by waiting (radius 1.5 A: -5 % when
            // every slab is paced, +1 % with the limit at 48; 2.0 A: +2 %). Waiting for the stores' acknowledgement
            // instead (s_waitcnt vmcnt(0)) gives +4 % where the sleep gives +4.7 %; a sleep after every store instruction,
            // or waves starting their first round apart, the same or less (profiles/r03_round_pacing.txt).
            if (rd < 3 && any == 2) __builtin_amdgcn_s_sleep(MVX_RSLEEP);
        };
        if (RUNS && !P.vec_store) { // rows that are not whole 16-byte quads: the tile holds the slab's runs as they lie in memory
            const RunLayout R = run_layout(NW, x0, y0, z0, P);
            const size_t S0 = (((size_t)b * P.C + L.cbase) * D + x0) * D2 + (size_t)y0 * D + z0;
            const int col = SUBZ * wave + lz;
            const bool zok = !R.joined || col < D;
            const int mine_r = 4 * h * R.SC + ly * R.SY + col; // + c * SC + x * SX
#pragma unroll
            for (int rd = 0; rd < NROUND; ++rd) {
                const size_t S0r = S0 + (size_t)(rd * CR) * D3;
                const int L0 = run_tile_origin(S0r, out);
                __syncthreads();
                if (zok) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        tile[L0 + mine_r + c * R.SC] = acc.p0[4 * rd + c];
                        tile[L0 + mine_r + c * R.SC + R.SX] = acc.p1[4 * rd + c];
                    }
                }
                __syncthreads();
                store_runs<false>(tile, R, L0, CR, L.cbase + rd * CR, S0r, tid, NW * 64, out, P);
            }
            return;
        }
        typedef std::integral_constant<int, 0> R0;
        typedef std::integral_constant<int, 1> R1;
        typedef std::integral_constant<int, 2> R2;
        typedef std::integral_constant<int, 3> R3;
        static_assert(NROUND == 4, "the rotation below spells out four rounds");
#ifdef MVX_ROT // (A/B builds) workgroups start their channel rounds at different channel groups - measured, no gain: by
               // blockIdx.x / 8: cfg-2 0.743 against 0.768, cfg-5 x 8 0.683 against 0.699; by blockIdx.x + y: 0.768 / 0.703
        switch ((MVX_ROT == 1 ? (blockIdx.x >> 3) : (blockIdx.x + blockIdx.y)) & 3u) {
        case 1: round(R1{}); round(R2{}); round(R3{}); round(R0{}); break;
        case 2: round(R2{}); round(R3{}); round(R0{}); round(R1{}); break;
        case 3: round(R3{}); round(R0{}); round(R1{}); round(R2{}); break;
        default: round(R0{}); round(R1{}); round(R2{}); round(R3{}); break;
        }
#else
        round(R0{});
        round(R1{});
        round(R2{});
        round(R3{});
#endif
    }
};

// ---- float64 grids, 32 channels, on the matrix cores ---------------------------------------------------------------------
// The same idea as OpsMx32 with v_mfma_f64_16x16x4_f64: D(16 x 16) += A(16 x 4) B(4 x 16), float64, the four k steps
// accumulated in sequence with one rounding each (tools/micro/mfma64_layout.hip) - bit for bit the chain of fma in
// candidate order of accumulate_row64. Operands one double per lane: A[i = lane % 16][k = lane / 16], B[k][j = lane % 16];
// D[i = 4 r + lane / 16][j = lane % 16] in register r = 0..3. Per FOUR candidates (k = lane / 16 picks the lane's
// candidate): lane l evaluates its candidate for four voxels - (x0 | x0 + 1, y0 + ly | y0 + 2 + ly, z) with (ly, lz) =
// ((l % 16) / 8, l % 8): dx^2 and dy^2 each shared by two of them, dz^2 by all four - and the 64 voxels x 32 channels take
// eight MFMAs (4 voxel blocks x 2 channel blocks, A = 16 channel weights of the candidate, B = the block's values).
// The vector path spent 32 float64 FMAs (128 issue cycles) and sixteen 16-byte LDS reads per candidate and wave on this.
typedef double d4v __attribute__((ext_vector_type(4)));
// exp(x) for x <= 0 in float64, for the matrix-core float64 walk: the library's exp is ~45 float64 instructions and, four
// per lane and candidate group, was what that walk spent most of its vector issue (and 27 spilled registers) on.
// x = (64 m + j) ln2 / 64 + r, |r| <= ln2 / 128: exp(x) = 2^m * 2^(j/64) * (1 + expm1(r)), the 64 table values correctly
// rounded (generated with 60-digit decimals), ln2 / 64 split so that k * LN2_64_HI is exact for |k| < 2^20, expm1 by its
// series to r^6 (the next term is below 3e-20 relative): ~1 ulp, like the library's (the reference's own chain of sqrt, two
// divisions and a square puts its argument ~4 ulp away already; the float64 goldens hold values to 1e-12).
__constant__ double EXP2_64TH[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
__device__ __forceinline__ double exp_nonpos64(double x, const double *__restrict__ tab /* LDS copy of EXP2_64TH */) {
    const double kd = __builtin_rint(x * 0x1.71547652b82fep+6); // 64 / ln2
    double r = fma(-kd, 0x1.62e42fee00000p-7, x);                // ln2 / 64, upper 32 bits
    r = fma(-kd, 0x1.a39ef35793c76p-39, r);                       // ... and the rest
    const int ki = (int)kd;
    const double t = tab[ki & 63];
    double q = fma(0x1.6c16c16c16c17p-10, r, 0x1.1111111111111p-7); // 1/720, 1/120
    q = fma(q, r, 0x1.5555555555555p-5);                             // 1/24
    q = fma(q, r, 0x1.5555555555555p-3);                             // 1/6
    q = fma(q, r, 0.5);
    const double p = fma(q * r, r, r); // expm1(r)
    return ldexp(fma(t, p, t), ki >> 6);
}
constexpr int MX64_SW = 84; // 16 + 64 words, padded to an odd number of 16-B quads (one row per lane in the row filter)
size_t voxelize_mx64_lds_bytes(int32_t NW) {
    const size_t rows = (size_t)64 * MX64_SW * 4;
    const size_t tile = (size_t)CR64 * RPC * row_stride_doubles(NW) * 8;
    return rows > tile ? rows : tile;
}

template <bool GAUSS, bool LANE_RANGE>
struct OpsMx64 {
    static constexpr bool RUNS = false;
    static constexpr int CT = 32;
    static constexpr bool GROUPED = false;
    struct Acc {
        d4v a[2][4]; // [channel block of 16][voxel block m = 2 x + yh]: channels 16 cb + 4 r + lane / 16, r = 0..3
    };
    static constexpr int WORDS = 2;
    static constexpr int WW = 64;
    static constexpr int SW = MX64_SW;
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc.a[cb][m] = (d4v){0.0, 0.0, 0.0, 0.0};
    }
    static __device__ __forceinline__ LaneCtx ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase, const VoxParams &P) {
        const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & 1;
        LaneCtx L;
        L.ix = x0;
        L.iy = y0 + ly;
        L.iz = z0 + SUBZ * wave + lz;
        L.gx = (double)L.ix * P.res - P.half;
        L.gx1 = (double)(L.ix + 1) * P.res - P.half;
        L.gy = (double)L.iy * P.res - P.half;
        L.gy1 = (double)(L.iy + 2) * P.res - P.half;
        L.gz = (double)L.iz * P.res - P.half;
        L.zt_w = zt_lo + wave;
        L.cbase = cbase;
        return L;
    }
    static __device__ __forceinline__ void walk(Acc &acc, unsigned long long mask, const unsigned *un, int lane, const LaneCtx &L,
                                                const VoxParams &P, const double *__restrict__, const float *__restrict__) {
        const int q = lane >> 4; // which candidate of the four this lane evaluates
        const int j = lane & 15; // ... and which of its 16-channel blocks' weights it feeds
        while (mask) {
            // up to four rows, in order (uniform)
            int s[4];
            int have = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] = 0;
                if (mask) {
                    s[k] = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    have = k + 1;
                }
            }
            const bool valid = q < have; // a short last group: the missing candidates add fma(0, 0, acc) = acc
            const int sq = q == 0 ? s[0] : (q == 1 ? s[1] : (q == 2 ? s[2] : s[3]));
            const unsigned *r = un + sq * SW;
            const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
            const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
            const double dx0 = Pxy.x - L.gx, dx1 = Pxy.x - L.gx1, dy0 = Pxy.y - L.gy, dy1 = Pxy.y - L.gy1, dz = PzT.x - L.gz;
            const double sx0 = dx0 * dx0, sx1 = dx1 * dx1, sy0 = dy0 * dy0, sy1 = dy1 * dy1, sz = dz * dz;
            double d2[4]; // m = 2 x + yh; cdist order (dx^2 + dy^2) + dz^2, no fma
            d2[0] = (sx0 + sy0) + sz;
            d2[1] = (sx0 + sy1) + sz;
            d2[2] = (sx1 + sy0) + sz;
            d2[3] = (sx1 + sy1) + sz;
            bool hit[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) hit[m] = valid && d2[m] <= PzT.y;
            if (LANE_RANGE) {
                const uint4 rg = *reinterpret_cast<const uint4 *>(r + 8); // k, type, xr, yr
                const unsigned zr = r[12];
                const bool zok = (L.iz >= (int)(zr & 0xffff)) && (L.iz <= (int)(zr >> 16));
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int ix = L.ix + (m >> 1), iy = L.iy + 2 * (m & 1);
                    hit[m] = hit[m] && zok && (ix >= (int)(rg.z & 0xffff)) && (ix <= (int)(rg.z >> 16)) && (iy >= (int)(rg.w & 0xffff)) &&
                             (iy <= (int)(rg.w >> 16));
                }
            }
            const double w0 = valid ? *reinterpret_cast<const double *>(r + 16 + 2 * j) : 0.0;        // channel j of the row
            const double w1 = valid ? *reinterpret_cast<const double *>(r + 16 + 2 * (16 + j)) : 0.0; // channel 16 + j
            const double c64 = GAUSS ? *reinterpret_cast<const double *>(r + 14) : 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) { // (one voxel block at a time: its value lives only until its two MFMAs are issued)
                double val = hit[m] ? 1.0 : 0.0;
                if (GAUSS && __ballot(hit[m]) != 0ull) { // (wave-uniform)
                    const double e = exp_nonpos64(c64 * d2[m], L.gtab);
                    val = hit[m] ? e : 0.0;
                }
                acc.a[0][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, val, acc.a[0][m], 0, 0, 0);
                acc.a[1][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, val, acc.a[1][m], 0, 0, 0);
            }
        }
    }
    // Write-out through the float64 tile of write_slab64 ([channel][x, y row][z], CR64 = 8 channels per round). Round t
    // holds channels 8 t .. 8 t + 7 = channel block t / 2, i = 8 (t % 2) + 4 rr + lane / 16 with rr = 0, 1, i.e. registers
    // r = 2 (t % 2) + rr of every lane: each lane writes 2 channels x 4 voxels per round.
    static __device__ __forceinline__ void write(const Acc &acc, int any, unsigned *un, int tid, int lane, int wave, int NW, int b,
                                                 const LaneCtx &L, int x0, int y0, int z0, void *out_, const VoxParams &P) {
        double *out = static_cast<double *>(out_);
        double *tile = reinterpret_cast<double *>(un);
        constexpr int CR = CR64, NROUND = 32 / CR;
        const int D = P.D;
        const int RS = row_stride_doubles(NW);
        const size_t D2 = (size_t)D * D, D3 = D2 * D;
        const int F2 = (SUBZ / 2) * NW; // 16-B slots per row
        const int nthr = NW * 64;
        const int rows_per_pass = nthr / F2; // 16 for any NW
        const int qq = tid % F2, rfirst = tid / F2;
        const int zq = z0 + 2 * qq;
        const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & 1, q = lane >> 4;
        const bool vec = P.vec_store != 0; // D even and a 16-B aligned grid
        double *mine = tile + (q * RPC + ly) * RS + SUBZ * wave + lz; // + (4 rr * RPC + x * SUBY + 2 yh) * RS
#pragma unroll
        for (int t = 0; t < NROUND; ++t) {
            __syncthreads(); // candidate rows (round 0) / previous tile (later rounds) fully consumed
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    mine[(4 * rr * RPC + (m >> 1) * SUBY + 2 * (m & 1)) * RS] = acc.a[t / 2][m][2 * (t % 2) + rr];
            __syncthreads();
            for (int row = rfirst; row < CR * RPC; row += rows_per_pass) {
                const int c = row / RPC, rw = row - c * RPC;
                const int sxx = (rw >> SUBY_SH) & (SUBX - 1), syy = rw & (SUBY - 1);
                const int ch = L.cbase + t * CR + c;
                if (ch < P.C && x0 + sxx < D && y0 + syy < D && zq < D) {
                    const double2 v = *reinterpret_cast<const double2 *>(tile + row * RS + 2 * qq);
                    double *dst = out + ((size_t)b * P.C + ch) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
                    if (vec) {
                        typedef double d2v __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store((d2v){v.x, v.y}, reinterpret_cast<d2v *>(dst));
                    } else {
                        dst[0] = v.x;
                        if (zq + 1 < D) dst[1] = v.y;
                    }
                }
            }
            // (round pacing as in OpsMx32::write, measured here: -7 % at 1280 cycles per round, -11 % at 2560 - two
            // workgroups per unit with a float64 walk between their write-outs do not queue loads behind stores)
            (void)any;
        }
    }
};

template <int CT_, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
struct OpsF64 {
    static constexpr bool RUNS = false;
    static constexpr int CT = CT_;
    typedef double Acc[CT];
    static constexpr int WORDS = 2;
    static constexpr int WW = 2 * CT;
    static constexpr int SW = 16 + 2 * CT;
    static_assert(16 + WW <= 128, "a row is staged by at most two wave-wide loads");
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.0;
    }
    static __device__ __forceinline__ void accumulate(Acc &acc, const unsigned *r, const LaneCtx &L, const VoxParams &P,
                                                      const double *__restrict__ Tc, const float *__restrict__ kc) {
        // (float64 handles keep float64 per-channel coefficients behind the `kc` pointer)
        accumulate_row64<CT, GAUSS, CHANWISE, LANE_RANGE>(acc, r, L, P.C, Tc, reinterpret_cast<const double *>(kc));
    }
    static __device__ __forceinline__ void write(const Acc &acc, bool, unsigned *un, int tid, int lane, int wave, int NW, int b,
                                                 const LaneCtx &L, int x0, int y0, int z0, void *out, const VoxParams &P) {
        write_slab64<CT>(acc, reinterpret_cast<double *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0,
                         static_cast<double *>(out), P);
    }
};

// One round of the line path: entries [e0, e0 + RW) of a slab line sit in the lanes of Er (lane l = entry e0 + l);
// candidates are entries 1..n_line. Stages their rows (slot = lane index) and walks them. Ends without a barrier.
template <typename Ops>
__device__ __forceinline__ void line_round(typename Ops::Acc &acc, const uint2 Er, int e0, int n_line, int RW,
                                           unsigned *un, const unsigned *__restrict__ rec, const unsigned *__restrict__ w,
                                           int64_t a0, int lane, int wave,
                                           int NW, const LaneCtx &L, const VoxParams &P, const double *__restrict__ Tc,
                                           const float *__restrict__ kc) {
    constexpr int SW = Ops::SW;
    // lanes 0-15 fetch the record, lanes 16.. the channel weights of the chunk: one load instruction per row
    const unsigned *src = lane < 16 ? rec + lane : w + (Ops::WORDS * L.cbase + lane - 16);
    const size_t stride = lane < 16 ? (size_t)16 : (size_t)(Ops::WORDS * P.w_stride);
    const bool stager = lane < 16 + Ops::WW;
    // rows wider than a wave (float64, 32 channels: 16 + 64 words): lanes 0.. fetch words 64.. with a second load
    constexpr int TAIL = 16 + Ops::WW > 64 ? 16 + Ops::WW - 64 : 0;
    const unsigned *src2 = w + (Ops::WORDS * L.cbase + lane + 48);
    unsigned v[8], v2[TAIL ? 8 : 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = wave + u * NW; // row slot staged by this wave (wave-uniform) <-> entry e0 + sl
        v[u] = 0u;
        if (TAIL) v2[u] = 0u;
        if (sl < RW && e0 + sl >= 1 && e0 + sl <= n_line) {
            const int ai = __builtin_amdgcn_readlane((int)Er.x, sl & 63);
            if (stager) v[u] = src[(size_t)(a0 + ai) * stride];
            if (TAIL && lane < TAIL) v2[u] = src2[(size_t)(a0 + ai) * (size_t)(Ops::WORDS * P.w_stride)];
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = wave + u * NW;
        if (sl < RW && e0 + sl >= 1 && e0 + sl <= n_line && stager) un[sl * SW + lane] = v[u];
        if (TAIL && sl < RW && e0 + sl >= 1 && e0 + sl <= n_line && lane < TAIL) un[sl * SW + 64 + lane] = v2[u];
    }
    VK_STAMP(2);
    __syncthreads();
    VK_STAMP(3);
    const unsigned pk = Er.y;
    const bool ok = (lane < RW) && (e0 + lane >= 1) && (e0 + lane <= n_line) && ((int)((pk >> 16) & 0xff) <= L.zt_w) &&
                    ((int)(pk >> 24) >= L.zt_w);
    unsigned long long mask = __ballot(ok);
    while (mask) {
        const int sl = __builtin_ctzll(mask);
        mask &= mask - 1;
        Ops::accumulate(acc, un + sl * SW, L, P, Tc, kc);
    }
    VK_STAMP(8 + wave); // every wave's own walk end
}

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
    size_t stride;       // words between the rows of consecutive atoms, for this lane
    bool stager;         // this lane takes part in staging
    const unsigned *src2; // rows wider than a wave (float64, 32 channels: 16 + 64 words): lanes 0.. fetch words 64.. too
    size_t stride2;
};
template <typename Ops>
constexpr int round_tail() { return 16 + Ops::WW > 64 ? 16 + Ops::WW - 64 : 0; }
template <typename Ops>
__device__ __forceinline__ RoundSrc<Ops> round_src(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, int lane,
                                                   const LaneCtx &L, const VoxParams &P) {
    RoundSrc<Ops> R;
    R.src = lane < 16 ? rec + lane : w + (Ops::WORDS * L.cbase + lane - 16);
    R.stride = lane < 16 ? (size_t)16 : (size_t)(Ops::WORDS * P.w_stride);
    // (grouped launches read the caller's feature rows in place whatever C is: no column beyond the row)
    R.stager = lane < 16 + Ops::WW && (!Ops::GROUPED || lane < 16 || L.cbase + lane - 16 < P.C);
    R.src2 = w + (Ops::WORDS * L.cbase + lane + 48);
    R.stride2 = (size_t)(Ops::WORDS * P.w_stride);
    return R;
}

// Stage a round through registers: eight scalar loads of atom indices, eight row loads in flight, eight LDS writes.
template <typename Ops>
__device__ __forceinline__ void stage_round(const uint2 *__restrict__ line, const uint2 *__restrict__ ext, int e0, int n_line,
                                            unsigned *un, const RoundSrc<Ops> &R, int64_t a0, int lane, int wave, int NW) {
    constexpr int SW = Ops::SW;
    int ai[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { // eight independent scalar loads
        const int e = e0 + wave + u * NW;
        ai[u] = (e >= 1 && e <= n_line) ? (int)(e < SLOTS ? line[e].x : ext[e - SLOTS].x) : 0;
    }
    constexpr int TAIL = round_tail<Ops>();
    unsigned v[8], v2[TAIL ? 8 : 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = e0 + wave + u * NW;
        v[u] = 0u;
        if (TAIL) v2[u] = 0u;
        if (e >= 1 && e <= n_line && R.stager) v[u] = R.src[(size_t)(a0 + ai[u]) * R.stride];
        if (TAIL && e >= 1 && e <= n_line && lane < TAIL) v2[u] = R.src2[(size_t)(a0 + ai[u]) * R.stride2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = wave + u * NW, e = e0 + sl;
        if (e >= 1 && e <= n_line && (R.stager || (Ops::GROUPED && lane < 16 + Ops::WW))) un[sl * SW + lane] = v[u]; // (v = 0 beyond C)
        if (TAIL && e >= 1 && e <= n_line && lane < TAIL) un[sl * SW + 64 + lane] = v2[u];
    }
}

// The same rows by LDS-DMA (global_load_lds_dword: destination = wave-uniform row base + 4 * lane, no data registers):
// issued for round r + 1 BEFORE round r is walked, into the other row buffer, so that a dense slab's further rounds find
// their rows in LDS when the walk before them ends (phase timeline at a 2.0 A radius: a staging phase waits ~9 000 cycles
// for its row loads behind the resident workgroups' stores, as long as a walk takes). Waited for by the vmcnt(0) that
// __syncthreads() carries. (Weight lanes beyond C of a grouped launch are left as they are: their slot mask is empty.)
template <typename Ops>
__device__ __forceinline__ void prefetch_round(const uint2 *__restrict__ line, const uint2 *__restrict__ ext, int e0, int n_line,
                                               unsigned *un, const RoundSrc<Ops> &R, int64_t a0, int wave, int NW) {
    constexpr int SW = Ops::SW;
    typedef const void __attribute__((address_space(1))) *gptr_t;
    typedef void __attribute__((address_space(3))) *lptr_t;
    // The atom indices through the scalar path, by hand: once an LDS-DMA (a store, to the compiler) has been issued in
    // the loop, hipcc no longer treats the line as invariant and fetches these entries with global_load_dword - through
    // the vector memory pipeline, behind the stores, one exposed latency per row. (s_load by asm is outside hipcc's
    // s_waitcnt bookkeeping: the wait is part of the statement that hands the values over.)
    int ai[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = e0 + wave + u * NW;
        const bool in = e >= 1 && e <= n_line;
        const uint2 *at = in ? (e < SLOTS ? line + e : ext + (e - SLOTS)) : line;
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(ai[u]) : "s"(at) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(ai[0]), "+s"(ai[1]), "+s"(ai[2]), "+s"(ai[3]), "+s"(ai[4]), "+s"(ai[5]), "+s"(ai[6]), "+s"(ai[7])
                 :
                 : "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = wave + u * NW, e = e0 + sl;
        if (e >= 1 && e <= n_line && R.stager)
            __builtin_amdgcn_global_load_lds((gptr_t)(R.src + (size_t)(a0 + ai[u]) * R.stride), (lptr_t)(un + sl * SW), 4, 0, 0);
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
#ifndef MVX_NO_CULL // (A/B builds)
        ok = ok && reaches_subtile(r, lane, L, P);
#endif
        if (xl) {
            const unsigned yr = r[11];
            const int sy = L.iy >> SUBY_SH; // (the slab's y index: the same for every lane)
            ok = ok && ((int)((yr & 0xffff) >> SUBY_SH) <= sy) && ((int)((yr >> 16) >> SUBY_SH) >= sy);
        }
    }
    Ops::walk(acc, __ballot(ok), un, lane, L, P, Tc, kc);
    VK_STAMP(8 + wave); // every wave's own walk end
}

// voxelize_kernel's arithmetic: 32-channel chunks go to the matrix cores (OpsMx32), everything else to the vector ALU
template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, bool GROUPED>
struct SlabOps {
    typedef OpsF32<CT, GAUSS, CHANWISE, LANE_RANGE> type;
};
#ifndef MVX_NO_MX // (A/B builds)
// (not the per-lane-range variants - blockdim 4, 5, 12, ...: their six extra index comparisons per voxel do not fit the
// 64 registers of the two-voxel layout without scratch: 0.527 against 0.479 ms per 64 cfg-2 molecules at blockdim 5)
template <bool GAUSS, bool GROUPED>
struct SlabOps<32, GAUSS, false, false, GROUPED> {
    typedef OpsMx32<GAUSS, false, GROUPED> type;
};
// (grouped launches - channel-wise features by radius - exist on the matrix-core path only, per-lane ranges or not)
template <bool GAUSS>
struct SlabOps<32, GAUSS, false, true, true> {
    typedef OpsMx32<GAUSS, true, true> type;
};
#endif

#ifndef MVX_BIG_WPE // waves per SIMD the 1024-thread variants (slabs of 9 ... 16 waves: whole rows of 65 ... 128 voxels) are compiled for -
                    // 8: 64 registers, two or three workgroups per compute unit (D = 72 4.16 TB/s against 3.75 with 4: 128 registers, one)
#define MVX_BIG_WPE 8
#endif
template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT, bool GROUPED = false>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 8 : MVX_BIG_WPE))
    voxelize_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                    const uint2 *__restrict__ slist_ext, const double *__restrict__ Tc, const float *__restrict__ kc, float *__restrict__ out,
                    const VoxParams P) {
    typedef typename SlabOps<CT, GAUSS, CHANWISE, LANE_RANGE, GROUPED>::type Ops;
#include "mvx_slab_body.inc"
}

// 32-channel chunks of float32 grids whose rows are not whole 16-byte quads (odd dimensions, unaligned grid slices): the
// matrix-core walk with the run-wise write-out (store_runs). A kernel of its own: compiled into voxelize_kernel<32, ...>
// the extra write-out path costs the aligned-grid kernels six more spilled registers and 0.5 % of the headline rate.
template <bool GAUSS, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 8 : MVX_BIG_WPE))
    voxelize_runs_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                         const uint2 *__restrict__ slist_ext, const double *__restrict__ Tc, const float *__restrict__ kc,
                         float *__restrict__ out, const VoxParams P) {
    typedef OpsMx32<GAUSS, false, false, true> Ops;
    constexpr int CT = 32;
    constexpr bool CHANWISE = false, GROUPED = false;
#include "mvx_slab_body.inc"
}

// (Measured and removed, round 3: channel counts of 32 k + r with the remainder chunk's workgroups - CTR vector-ALU
// accumulators - in the SAME launch as the full chunks' (a kernel that branches on the chunk index into two inclusions of
// the slab body). Bit-identical, and slower than the second launch it replaced: C = 33 0.618 against 0.580 ms per 64
// molecules, C = 40 0.647 / 0.605, C = 48 0.708 / 0.652, C = 65 0.967 / 0.912 - the remainder's workgroups take slots from
// the store-streaming ones for longer than their own launch lasts.)

// float64 grids, chunks of 32 channels, scalar / atom-wise radii: the slab body with OpsMx64 (128 registers)
template <bool GAUSS, bool LANE_RANGE, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 4 : 2))
    voxelize64_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                      const uint2 *__restrict__ slist_ext, double *__restrict__ out, const VoxParams P) {
    typedef OpsMx64<GAUSS, LANE_RANGE> Ops;
    constexpr int CT = 32;
    constexpr bool CHANWISE = false, GROUPED = false;
    const double *const Tc = nullptr;
    const float *const kc = nullptr;
#include "mvx_slab_body.inc"
}

// The general slab loop (float64 grids that do not take voxelize64_kernel): all `total` (molecule, chunk, slab) ids,
// grid-stride. (Until round 3 it also served the float32 overflow list.)
template <typename Ops, int MAXT, int WPE>
__global__ void __launch_bounds__(MAXT, WPE)
    voxelize_dense_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ xlist,
                          const uint2 *__restrict__ slist, const uint2 *__restrict__ slist_ext,
                          const int64_t *__restrict__ offsets, int64_t n_one, const double *__restrict__ Tc,
                          const float *__restrict__ kc, void *__restrict__ out, const VoxParams P, unsigned T, unsigned total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SW = Ops::SW;
    constexpr int CT = Ops::CT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = P.NW;
    const int SB = NW < 4 ? NW : 4; // x-list entries per lane and scan round
    const int LCAP = 64 * SB;
    int *list = reinterpret_cast<int *>(smem);
    unsigned *zr_l = reinterpret_cast<unsigned *>(smem + 4 * LCAP);
    int *nlist_s = reinterpret_cast<int *>(smem + 8 * LCAP);
    unsigned *un = reinterpret_cast<unsigned *>(smem + 8 * LCAP + 16);
    const int RW = 8 * NW < 64 ? 8 * NW : 64;

    for (unsigned id = blockIdx.x; id < total; id += gridDim.x) {
        const unsigned z = id / T, t = id - z * T;
        int b = (int)z, cc = 0;
        if (P.ncc > 1) {
            b = (int)z / P.ncc;
            cc = (int)z - b * P.ncc;
        }
        int sx, sy, zc;
        decode_slab(t, P, sx, sy, zc);
        const int x0 = SUBX * sx, y0 = SUBY * sy, z0 = zc * SUBZ * NW;
        const int zt_lo = zc * NW, zt_hi = zt_lo + NW - 1;
        const LaneCtx L = make_lane_ctx(lane, wave, x0, y0, z0, zt_lo, cc * CT, P);
        const uint2 *__restrict__ line = slist + ((size_t)b * T + t) * SLOTS;
        const uint2 *__restrict__ ext = slist_ext + ((size_t)b * T + t) * EXT_SLOTS;
        const uint2 hdr = line[0];
        const int64_t a0 = (int64_t)hdr.y;
        typename Ops::Acc acc;
        Ops::zero(acc);
        bool any = false;

        if (hdr.x != LINE_OVERFLOW) {
            // rounds of RW entries over the primary line and its extension
            const int n_line = (int)hdr.x;
            any = n_line > 0;
            for (int e0 = 0; e0 <= n_line && n_line > 0; e0 += RW) {
                if (e0 > 0) __syncthreads(); // rows of the previous round consumed
                const int e = e0 + lane;
                uint2 Er = make_uint2(0u, EMPTY_ENTRY);
                if (lane < RW && e <= n_line) Er = (e < SLOTS) ? line[e] : ext[e - SLOTS];
                line_round<Ops>(acc, Er, e0, n_line, RW, un, rec, w, a0, lane, wave, NW, L, P, Tc, kc);
            }
        } else {
            // x-list path: more candidates than a line and its extension hold
            const int64_t nmol = offsets ? offsets[b + 1] - offsets[b] : n_one;
            const uint2 *__restrict__ xl = xlist + ((size_t)a0 + 2 * (size_t)b) * P.nsx + (size_t)sx * (size_t)(nmol + XL_HEADER);
            const int nx = (int)xl[0].x + XL_HEADER;
            const unsigned *src = lane < 16 ? rec + lane : w + (Ops::WORDS * L.cbase + lane - 16);
            const size_t stride = lane < 16 ? (size_t)16 : (size_t)(Ops::WORDS * P.w_stride);
            const bool stager = lane < 16 + Ops::WW;
            constexpr int TAIL = 16 + Ops::WW > 64 ? 16 + Ops::WW - 64 : 0;
            for (int base = 0; base < nx; base += LCAP) {
                __syncthreads(); // list / candidate rows of the previous round consumed
                if (wave == 0) { // ordered compaction of LCAP entries against the slab's y/z box
                    int n = 0;
                    for (int u = 0; u < SB; ++u) {
                        const int i = base + u * 64 + lane;
                        const uint2 en = (i < nx) ? xl[i] : make_uint2(0u, EMPTY_ENTRY);
                        // (the two header entries carry EMPTY_ENTRY and never match)
                        const bool m = ((int)(en.y & 0xff) <= sy) && ((int)((en.y >> 8) & 0xff) >= sy) &&
                                       ((int)((en.y >> 16) & 0xff) <= zt_hi) && ((int)(en.y >> 24) >= zt_lo);
                        const unsigned long long mask = __ballot(m);
                        if (m) {
                            const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            list[pos] = (int)en.x;
                            zr_l[pos] = en.y;
                        }
                        n += __popcll(mask);
                    }
                    if (lane == 0) nlist_s[0] = n;
                }
                __syncthreads();
                const int nl = nlist_s[0];
                any = any || nl > 0;
                for (int c0 = 0; c0 < nl; c0 += P.dcap) {
                    const int n = (nl - c0) < P.dcap ? (nl - c0) : P.dcap; // rows staged this round
                    if (c0 > 0) __syncthreads();
                    for (int j = wave; j < n; j += NW) {
                        if (stager) un[j * SW + lane] = src[(size_t)(a0 + list[c0 + j]) * stride];
                        if (TAIL && lane < TAIL) // (rows wider than a wave: float64, 32 channels)
                            un[j * SW + 64 + lane] = (w + (Ops::WORDS * L.cbase + lane + 48))[(size_t)(a0 + list[c0 + j]) * (size_t)(Ops::WORDS * P.w_stride)];
                    }
                    __syncthreads();
                    for (int jb = 0; jb < n; jb += 64) {
                        const int j = jb + lane;
                        bool ok = false;
                        if (j < n) {
                            const unsigned zr = zr_l[c0 + j];
                            ok = ((int)((zr >> 16) & 0xff) <= L.zt_w) && ((int)(zr >> 24) >= L.zt_w);
                        }
                        unsigned long long mask = __ballot(ok);
                        while (mask) {
                            const int jj = jb + __builtin_ctzll(mask);
                            mask &= mask - 1;
                            Ops::accumulate(acc, un + jj * SW, L, P, Tc, kc);
                        }
                    }
                }
            }
        }
        Ops::write(acc, any, un, tid, lane, wave, NW, b, L, x0, y0, z0, out, P);
        __syncthreads(); // rows / tile consumed before the next slab's rows land in the union region
    }
}

// ------------------------------------------------------------------------------------------------
// voxelize_direct_kernel: the whole call in ONE launch (per-molecule forward() calls, small batches)
// ------------------------------------------------------------------------------------------------
// The reference's unit of work is one molecule per forward() call (test/test_time_numpy.py:11-15). For such calls the
// three-launch pipeline above (prep -> xbin -> voxelize, + an H2D copy of the transform) is all latency: 25-40 us of
// launches and boundaries around 5-12 us of voxelize work. This kernel needs no workspace and no pre-pass:
//   grid = (slab, molecule * ncc + channel chunk) as voxelize_kernel. Per workgroup:
//   A. scan: wave w takes atoms [w*512, (w+1)*512) of the current 512*NW-atom segment, 128 at a time: the 3 KB of
//      coordinates are fetched with contiguous 16-B-per-lane loads (the (N,3) rows are 24 B apart: one load per
//      coordinate would touch every cache line three times, and every workgroup of the chip reads the same lines),
//      rounded to float32 and transposed through the wave's own LDS strip; the transform is applied in float32 and
//      the atom's radius window, widened by the float32 error bound (SCAN_MARGIN), is tested against the slab's box:
//      a superset of the atoms that can reach the slab (the exact float64 decisions are step B's). Survivors are appended, by ballot + prefix, to the wave's region of an LDS
//      list; one barrier. Regions in wave order = candidates IN ATOM ORDER (sums bit-identical to the binned path's).
//   B. rounds of up to 64 candidates: lane u < 8 of wave w prepares slot w + u*NW - position, exact box cull and the
//      cull of the reference blocks this slab lies in (x, y), threshold T, coefficient k - and writes the 64-B record
//      straight into the LDS row, while the wave's other lanes fetch the slot's channel weights from the caller's
//      feature rows (or build the one-hot / unit row of forward_types / forward_single); one barrier; every wave
//      then selects, one lane per row, the rows that pass ITS sub-tile's z block cull and z window, and walks them
//      (Ops::accumulate); write-out as everywhere (Ops::write).
//      Sub-tiles that straddle reference blocks (LANE_RANGE: blockdim 4, 5, 12, ...) need per-lane voxel ranges:
//      those variants run prep_atom per candidate instead, as the prep kernel does.
//   Any number of candidates and atoms works (rounds, segments); there is no overflow list and no dense kernel.
// LDS map: u16 list[NW*512] | int wcnt[16] | u32 pk[64] | double Tc[32] | float kc[32] | union { scan strips ; rows ; tile }.
constexpr int SEGW = 512;            // atoms one wave scans per segment
constexpr int SCAN_BLOCK = 128;      // atoms per coalesced fetch (3 x 1 KB)
constexpr int DIRECT_HDR_BYTES = 64 + 256 + 32 * 8 + 32 * 4; // wcnt + pk + Tc + kc

size_t direct_lds_bytes(int32_t ct, int32_t NW) {
    const size_t strips = (size_t)NW * SCAN_BLOCK * 12;
    const size_t un = voxelize_lds_bytes(ct, NW, DIRECT_CR);
    return (size_t)NW * SEGW * 2 + DIRECT_HDR_BYTES + (un > strips ? un : strips);
}

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
__device__ __forceinline__ bool block_admits(const BlockBounds &B, double p, double r) {
    return (!B.has_lo || p > B.lo - r) && (!B.has_hi || p < B.hi + r);
}

#ifndef MVX_DIRECT_WPS
#define MVX_DIRECT_WPS 4
#endif
template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT>
__global__ void __launch_bounds__(MAXT, MVX_DIRECT_WPS)
    voxelize_direct_kernel(const DirectArgs A, float *__restrict__ out, const VoxParams P) {
    typedef OpsF32<CT, GAUSS, CHANWISE, LANE_RANGE> Ops;
    constexpr int SW = Ops::SW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = P.NW;
    unsigned short *list = reinterpret_cast<unsigned short *>(smem);
    int *wcnt = reinterpret_cast<int *>(smem + (size_t)NW * SEGW * 2);
    unsigned *pk = reinterpret_cast<unsigned *>(wcnt + 16);
    double *Tc_s = reinterpret_cast<double *>(pk + 64);
    float *kc_s = reinterpret_cast<float *>(Tc_s + 32);
    unsigned *un = reinterpret_cast<unsigned *>(kc_s + 32);
    float *strip = reinterpret_cast<float *>(un) + (size_t)wave * SCAN_BLOCK * 3; // this wave's scan strip

    const unsigned t = blockIdx.x;
    int b = (int)blockIdx.y, cc = 0;
    if (P.ncc > 1) {
        b = (int)blockIdx.y / P.ncc;
        cc = (int)blockIdx.y - b * P.ncc;
    }
    int sx, sy, zc;
    decode_slab(t, P, sx, sy, zc);
    const int x0 = SUBX * sx, y0 = SUBY * sy, z0 = zc * SUBZ * NW;
    const int zt_lo = zc * NW, zt_hi = zt_lo + NW - 1;
    const int cbase = cc * CT; // first channel of this workgroup's chunk
    const PrepArgs &pa = A.pa;
    const int C = pa.C;
    const Geom &g = pa.g;

    int64_t a0 = 0, a1 = A.N;
    if (pa.offsets) {
        a0 = pa.offsets[b];
        a1 = pa.offsets[b + 1];
    }
    mvx_xform xf = pa.xf_one;
    if (pa.xforms) xf = pa.xforms[b];
    const bool has_xf = xf.flags != 0;

    // channel-wise features: per-channel thresholds / coefficients of this chunk and max(radii), as chan_aux_kernel
    float rmax32 = 0.0f;
    if constexpr (CHANWISE) {
        const float *cr = static_cast<const float *>(pa.radii);
        if (tid < CT) {
            const int ch = (cbase + tid < C) ? cbase + tid : C - 1;
            const float r = cr[ch];
            Tc_s[tid] = d2_threshold(r);
            kc_s[tid] = GAUSS ? gauss_coeff(r, pa.sigma32) : 0.0f;
        }
        float m = cr[0];
        for (int c = lane; c < C; c += 64) m = cr[c] > m ? cr[c] : m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
        rmax32 = m;
    }

    // the slab's box, widened per atom by its radius window (superset of the voxels the atom can reach)
    const int xh = (x0 + SUBX - 1 < P.D - 1) ? x0 + SUBX - 1 : P.D - 1;
    const int yh = (y0 + SUBY - 1 < P.D - 1) ? y0 + SUBY - 1 : P.D - 1;
    const int zh = (z0 + SUBZ * NW - 1 < P.D - 1) ? z0 + SUBZ * NW - 1 : P.D - 1;
    const double slack = 1e-6 * P.res;
    const double bx0 = uniform((double)x0 * P.res - P.half - slack), bx1 = uniform((double)xh * P.res - P.half + slack);
    const double by0 = uniform((double)y0 * P.res - P.half - slack), by1 = uniform((double)yh * P.res - P.half + slack);
    const double bz0 = uniform((double)z0 * P.res - P.half - slack), bz1 = uniform((double)zh * P.res - P.half + slack);
    const XformF32 X32 = make_xform_f32(xf);
    // the box as centre (minus the transform's final offset) and half extents, rounded outwards
    const float ccx = uniform((float)(0.5 * (bx0 + bx1)) - X32.o0), ccy = uniform((float)(0.5 * (by0 + by1)) - X32.o1),
                ccz = uniform((float)(0.5 * (bz0 + bz1)) - X32.o2);
    const float hx = uniform((float)(0.5 * (bx1 - bx0)) * 1.000001f + 1e-6f), hy = uniform((float)(0.5 * (by1 - by0)) * 1.000001f + 1e-6f),
                hz = uniform((float)(0.5 * (bz1 - bz0)) * 1.000001f + 1e-6f);

    const BlockBounds Bx = block_bounds(g, x0), By = block_bounds(g, y0); // the reference blocks this slab lies in
    const int RW = 8 * NW < 64 ? 8 * NW : 64; // candidate rows per round
    const int64_t SEGN = (int64_t)NW * SEGW;
    // a molecule that fits one round of rows (ligands) skips the scan: every atom is staged, and the stage's own
    // x / y window test drops the ones that cannot reach this slab
    const bool small = (a1 - a0) <= RW;
    if (tid < 16) wcnt[tid] = 0; // (wave 0, before its own count is stored; the scan's barrier publishes both)
#ifdef MVX_DIAG // per-workgroup s_memtime stamps into the (otherwise unused) record buffer: diagnostic builds only
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(pa.rec) + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#define MVX_STAMP(i) do { if (tid == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
    if (tid == 0) for (int i = 0; i < 8; ++i) stamps[i] = 0;
#else
#define MVX_STAMP(i) do { } while (0)
#endif
    MVX_STAMP(0);
#ifdef MVX_DIAG // run-time ablations (no dead-code elimination, same register allocation): 2 = no atoms at all
    if (P.dbg & 2) a1 = a0;
#endif

    // ---- A. scan of one segment: survivors of wave w -> list[w*SEGW ...], counts -> wcnt; returns their number and
    //         the exclusive prefixes of the NW counts (scalar registers) ---------------------------------------------
    auto scan = [&](int64_t seg0, int (&pre)[9]) -> int {
    const int64_t wbeg = seg0 + (int64_t)wave * SEGW;
    int cnt = 0;
    if (!small && wbeg < a1) {
        const int64_t dend = 3 * a1; // doubles of this molecule end here
        // the next 128-atom block's six loads are in flight while this block is tested (the loaded doubles stay
        // untouched in registers until the next iteration: converting them in the fetch would wait for them on the
        // spot); lane l holds doubles 128k + 2l, 128k + 2l + 1 (k = 0..2) of the 384-double block
        constexpr int NBLK = SEGW / SCAN_BLOCK;
        double fd[6];
        float fr[SCAN_BLOCK / 64]; // atom-wise radii of the block, fetched with it (a load inside the test loop would
                                   // be waited for with vmcnt(0), i.e. together with the whole prefetch)
        auto fetch = [&](int blk) {
            const int64_t d0 = 3 * (wbeg + (int64_t)blk * SCAN_BLOCK) + 2 * lane;
#pragma unroll
            for (int k = 0; k < 3; ++k) { // (clamped addresses, unconditional loads; values past the molecule are never used)
                const int64_t d = d0 + 128 * k;
                fd[2 * k] = pa.coords[d < dend ? d : dend - 1];
                fd[2 * k + 1] = pa.coords[d + 1 < dend ? d + 1 : dend - 1];
            }
            if (pa.radii_src == RAD_ATOM) {
#pragma unroll
                for (int q = 0; q < SCAN_BLOCK / 64; ++q) {
                    const int64_t a = wbeg + (int64_t)blk * SCAN_BLOCK + 64 * q + lane;
                    fr[q] = static_cast<const float *>(pa.radii)[a < a1 ? a : a1 - 1];
                }
            }
        };
        // one copy of the loop per kind of radius (one value for every atom / a load per atom)
        auto region = [&](auto per_atom_radius) {
            constexpr bool PER_ATOM = decltype(per_atom_radius)::value;
            float rscalar = 0.0f;
            if (pa.radii_src == RAD_SCALAR) rscalar = (float)pa.radius_scalar;
            else if (pa.radii_src == RAD_CHANNEL_FEATURES) rscalar = rmax32;
            fetch(0);
#pragma nounroll
            for (int blk = 0; blk < NBLK; ++blk) {
                if (wbeg + blk * SCAN_BLOCK >= a1) break;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    *reinterpret_cast<float2v *>(strip + 128 * k + 2 * lane) = (float2v){(float)fd[2 * k], (float)fd[2 * k + 1]};
                float rblk[SCAN_BLOCK / 64];
#pragma unroll
                for (int q = 0; q < SCAN_BLOCK / 64; ++q) rblk[q] = PER_ATOM ? fr[q] : 0.0f;
                if (blk + 1 < NBLK && wbeg + (blk + 1) * SCAN_BLOCK < a1) fetch(blk + 1);
#pragma unroll
                for (int q = 0; q < SCAN_BLOCK / 64; ++q) {
                    const int j = 64 * q + lane;
                    const int64_t a = wbeg + blk * SCAN_BLOCK + j;
                    bool ok = a < a1;
                    float x = strip[3 * j], y = strip[3 * j + 1], z = strip[3 * j + 2];
                    float rwin = rscalar;
                    if constexpr (PER_ATOM) {
                        if (pa.radii_src == RAD_ATOM) rwin = rblk[q];
                        else {
                            const int ty = pa.types[ok ? a : a1 - 1];
                            ok = ok & (ty >= 0) & (ty < C);
                            rwin = static_cast<const float *>(pa.radii)[ok ? ty : 0];
                        }
                    }
                    // float32 estimate of the position; every test widened by the estimate's error bound
                    const float mag = X32.scale * (fabsf(x) + fabsf(y) + fabsf(z)) + X32.mag;
                    x -= X32.c0;
                    y -= X32.c1;
                    z -= X32.c2;
                    if (X32.rot) {
                        const float u = X32.m00 * x + X32.m01 * y + X32.m02 * z;
                        const float v = X32.m10 * x + X32.m11 * y + X32.m12 * z;
                        const float w = X32.m20 * x + X32.m21 * y + X32.m22 * z;
                        x = u;
                        y = v;
                        z = w;
                    }
                    const float rr = rwin * 1.00001f + SCAN_MARGIN * mag + 1e-6f;
                    const bool near = (fabsf(x - ccx) <= hx + rr) & (fabsf(y - ccy) <= hy + rr) & (fabsf(z - ccz) <= hz + rr);
                    ok = ok & (near | !(mag < 1.0e30f)); // magnitudes float32 cannot hold: leave it to the float64 step
                    const unsigned long long mk = __ballot(ok);
                    if (ok)
                        list[wave * SEGW + cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] =
                            (unsigned short)(wave * SEGW + blk * SCAN_BLOCK + j);
                    cnt += __popcll(mk);
                }
            }
        };
        if (pa.radii_src == RAD_SCALAR || pa.radii_src == RAD_CHANNEL_FEATURES) region(std::false_type{});
        else region(std::true_type{});
    }
    pre[0] = 0;
    int total;
    if (small) { // every atom is a candidate: no scan, no list, no barrier
        total = (int)(a1 - a0);
#pragma unroll
        for (int w = 0; w < 8; ++w) pre[w + 1] = total;
    } else {
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        // the NW counts as exclusive prefixes in scalar registers (eight independent LDS reads, once per segment)
#pragma unroll
        for (int w = 0; w < 8; ++w) pre[w + 1] = pre[w] + __builtin_amdgcn_readfirstlane(wcnt[w]); // (entries >= NW stay 0)
        total = pre[8];
    }
        MVX_STAMP(1);
#ifdef MVX_DIAG // 1 = scan, but pretend nothing survived
        if (P.dbg & 1) total = 0;
#endif
        return total;
    };
    // candidate j of the segment (atom order) -> atom index inside the segment
    auto candidate = [&](const int (&pre)[9], int j) -> int {
        if (small) return j;
        int w = 0, base = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const bool ge = j >= pre[q];
            w = ge ? q : w;
            base = ge ? pre[q] : base;
        }
        return (int)list[w * SEGW + (j - base)];
    };
    // ---- B1. stage candidates [c0, c0 + n) of the segment: records + channel weights -> LDS rows; ends with a barrier
    auto stage = [&](int64_t seg0, const int (&pre)[9], int c0, int n) {
    // the atoms of this wave's slots: lane u < 8 <-> slot wave + u*NW (one list read for all eight)
    int my_idx = 0;
    {
        const int sl = wave + lane * NW;
        if (lane < 8 && sl < n) my_idx = candidate(pre, c0 + sl);
    }
    // B1. channel weights of those slots (atom indices broadcast by v_readlane), all loads in flight
    unsigned v[8];
    if (pa.mode == MODE_FEATURES) {
        const float *feat = static_cast<const float *>(pa.features);
        const bool wl = lane >= 16 && lane < 16 + CT && (cbase + lane - 16) < C;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = wave + u * NW;
            v[u] = 0u;
            if (sl < n) {
                const int64_t a = seg0 + (int64_t)__builtin_amdgcn_readlane(my_idx, u);
                if (wl) v[u] = __float_as_uint(feat[a * C + cbase + lane - 16]);
            }
        }
    }
    // the records of this wave's slots
    int my_type = 0;
    {
        const int sl = wave + lane * NW;
        if (lane < 8 && sl < n) {
            const int64_t a = seg0 + (int64_t)my_idx;
            double p[3] = {pa.coords[3 * a], pa.coords[3 * a + 1], pa.coords[3 * a + 2]};
#ifdef MVX_DIAG
            if (p[0] != 1.2345e300) MVX_STAMP(7); // (after the coordinates have arrived)
#endif
            if (has_xf) apply_xform(xf, p[0], p[1], p[2]);
            if constexpr (LANE_RANGE) {
                // per-lane voxel ranges are needed: the prep kernel's own code, one lane per candidate
                AtomRec R;
                uint32_t rng[3];
                bool keep = prep_atom(pa, a, p, rmax32, 0.0, R, rng);
                my_type = R.type;
                const int xlo = (int)(rng[0] & 0xffff), xhi = (int)(rng[0] >> 16);
                const int ylo = (int)(rng[1] & 0xffff) >> SUBY_SH, yhi = (int)(rng[1] >> 16) >> SUBY_SH;
                const int zlo = (int)(rng[2] & 0xffff) >> SUBZ_SH, zhi = (int)(rng[2] >> 16) >> SUBZ_SH;
                keep = keep && (xlo <= x0 + SUBX - 1) && (xhi >= x0) && (ylo <= sy) && (yhi >= sy) && (zlo <= zt_hi) &&
                       (zhi >= zt_lo);
                const uint4 *src = reinterpret_cast<const uint4 *>(&R);
                uint4 *dst = reinterpret_cast<uint4 *>(un + sl * SW);
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[i] = src[i];
                pk[sl] = keep ? (((unsigned)zlo << 16) | ((unsigned)zhi << 24)) : EMPTY_ENTRY;
            } else {
                // sub-tiles lie inside one reference block: the culls are uniform over this slab (x, y) and
                // over each wave's sub-tile (z, tested by the waves below); same comparisons as prep_atom
                const double ub = g.half, lb = -1 * g.half;
                float r32;
                double rc;
                bool keep = true;
                if (pa.types) {
                    my_type = pa.types[a];
                    if (my_type < 0 || my_type >= C) keep = false;
                }
                if (pa.radii_src == RAD_SCALAR) {
                    rc = pa.radius_scalar;
                    r32 = (float)pa.radius_scalar;
                    for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lb - rc) && (p[i] < ub + rc); // numpy/voxelizer.py:487-488
                } else if (pa.radii_src == RAD_CHANNEL_FEATURES) {
                    r32 = rmax32;
                    rc = (double)rmax32;
                    const double lo = (double)((float)lb - rmax32), hi = (double)((float)ub + rmax32); // NEP 50, :138
                    for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lo) && (p[i] < hi);
                } else {
                    const int64_t ri = (pa.radii_src == RAD_ATOM) ? a : (keep ? (int64_t)my_type : -1); // :284-285
                    r32 = ri >= 0 ? static_cast<const float *>(pa.radii)[ri] : 0.0f;
                    rc = (double)r32;
                    for (int i = 0; i < 3; ++i) keep = keep && (p[i] + rc > lb) && (p[i] - rc < ub); // :491-492
                }
                // (one python float for every atom: threshold and coefficient come with the launch)
                const double T = pa.radii_src == RAD_SCALAR ? pa.T_scalar : d2_threshold(r32);
                keep = keep && (T >= 0.0);
                keep = keep && block_admits(Bx, p[0], rc) && block_admits(By, p[1], rc);
                const double rrd = (double)r32 * 1.000001 + 1e-9; // conservative window, as prep_atom's
                keep = keep && (p[0] + rrd >= bx0) && (p[0] - rrd <= bx1) && (p[1] + rrd >= by0) && (p[1] - rrd <= by1);
                typedef double d2v __attribute__((ext_vector_type(2)));
                d2v *dst = reinterpret_cast<d2v *>(un + sl * SW);
                dst[0] = (d2v){p[0], p[1]};
                dst[1] = (d2v){p[2], T};
                un[sl * SW + 8] = __float_as_uint(!GAUSS ? 0.0f : (pa.radii_src == RAD_SCALAR ? pa.k_scalar : gauss_coeff(r32, pa.sigma32)));
                un[sl * SW + 9] = (unsigned)my_type;
                *reinterpret_cast<double *>(un + sl * SW + 10) = rc;
                // z window radius, rounded up to float; a dropped candidate gets a negative one
                un[sl * SW + 12] = __float_as_uint(keep ? (float)rrd * 1.0000002f : -1.0f);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sl = wave + u * NW;
        if (sl < n && lane >= 16 && lane < 16 + (CT < 4 ? 4 : CT)) {
            unsigned w;
            if (pa.mode == MODE_FEATURES) w = v[u];
            else if (pa.mode == MODE_TYPES) w = (__builtin_amdgcn_readlane(my_type, u) == cbase + lane - 16) ? 0x3f800000u : 0u;
            else w = (lane == 16) ? 0x3f800000u : 0u;
            un[sl * SW + lane] = w;
        }
    }
    MVX_STAMP(2);
    __syncthreads();
    MVX_STAMP(3);
    };

    // The first round of the first segment is staged BEFORE the accumulators exist: scan and stage then have the
    // whole register file (with the accumulators live their loops spill, and a scratch reload inside the scan loop
    // costs a full vmcnt(0) drain per block), and per-molecule calls rarely need more than this one round per slab.
    // Later rounds / segments run the same code with the accumulators live.
    int pre0[9];
    int total0 = 0;
    if (a1 > a0) {
        total0 = scan(a0, pre0);
        if (total0 > 0) stage(a0, pre0, 0, total0 < RW ? total0 : RW);
    }
    // (voxel centres and accumulators only from here on: see above)
    const LaneCtx L = make_lane_ctx(lane, wave, x0, y0, z0, zt_lo, cbase, P);
    typename Ops::Acc acc;
    Ops::zero(acc);
    bool any = false;
    // ---- B2. the rows this wave's sub-tile takes (one lane per row), then the walk; ends with a barrier ------------
    auto walk = [&](int n) {
    // B2. the rows this wave's sub-tile takes (one lane per row), then the walk
    {
        bool ok = false, kept = false;
        if (lane < n) {
            if constexpr (LANE_RANGE) {
                const unsigned pkl = pk[lane];
                kept = pkl != EMPTY_ENTRY;
                ok = ((int)((pkl >> 16) & 0xff) <= zt_lo + wave) && ((int)(pkl >> 24) >= zt_lo + wave);
            } else {
                const unsigned *r = un + lane * SW;
                const double pz = *reinterpret_cast<const double *>(r + 4);
                const double rc = *reinterpret_cast<const double *>(r + 10);
                const double rr = (double)__uint_as_float(r[12]);
                const int zv = z0 + SUBZ * wave; // first voxel of this wave's sub-tile
                const int zl = (zv + SUBZ - 1 < P.D - 1) ? zv + SUBZ - 1 : P.D - 1;
                const BlockBounds Bz = block_bounds(g, zv);
                kept = (rr >= 0.0) && (pz + rr >= bz0) && (pz - rr <= bz1);
                ok = kept && (zv < P.D) && block_admits(Bz, pz, rc) &&
                     (pz + rr >= (double)zv * P.res - P.half - slack) && (pz - rr <= (double)zl * P.res - P.half + slack);
            }
        }
        any = any || __ballot(kept) != 0ull; // (the same rows in every wave: workgroup-uniform)
        unsigned long long mask = __ballot(ok);
        while (mask) {
            const int sl = __builtin_ctzll(mask);
            mask &= mask - 1;
            Ops::accumulate(acc, un + sl * SW, L, P, Tc_s - cbase, kc_s - cbase);
        }
    }
    MVX_STAMP(4);
    __syncthreads(); // rows / pk consumed before the next round (or the next segment's scan strips) overwrite them
    };
    if (total0 > 0) {
        walk(total0 < RW ? total0 : RW);
#pragma nounroll
        for (int c0 = RW; c0 < total0; c0 += RW) {
            const int n = (total0 - c0) < RW ? (total0 - c0) : RW;
            stage(a0, pre0, c0, n);
            walk(n);
        }
    }
#pragma nounroll
    for (int64_t seg0 = a0 + SEGN; seg0 < a1; seg0 += SEGN) {
        int pre[9];
        const int total = scan(seg0, pre);
#pragma nounroll
        for (int c0 = 0; c0 < total; c0 += RW) {
            const int n = (total - c0) < RW ? (total - c0) : RW;
            stage(seg0, pre, c0, n);
            walk(n);
        }
    }
    MVX_STAMP(5);
    Ops::write_wide(acc, any, un, tid, lane, wave, NW, b, L, x0, y0, z0, out, P);
    MVX_STAMP(6);
#undef MVX_STAMP
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
struct KernelKey {
    int ct;
    bool gauss, chanwise, lane_range;
    int maxt;
};

// Calls fn.template operator()<CT, GAUSS, CHANWISE, LANE_RANGE, MAXT>() for the instantiation `k` names.
template <typename Fn>
static hipError_t for_kernel(const KernelKey &k, Fn &&fn) {
#define MVX_CASE(CT_, G_, CW_, LR_, MT_) \
    if (k.ct == CT_ && k.gauss == G_ && k.chanwise == CW_ && k.lane_range == LR_ && k.maxt == MT_) \
        return fn.template operator()<CT_, G_, CW_, LR_, MT_>();
#define MVX_CASES_CT(CT_, MT_)             \
    MVX_CASE(CT_, true, false, false, MT_)   \
    MVX_CASE(CT_, false, false, false, MT_)  \
    MVX_CASE(CT_, true, false, true, MT_)    \
    MVX_CASE(CT_, false, false, true, MT_)   \
    MVX_CASE(CT_, true, true, true, MT_)     \
    MVX_CASE(CT_, false, true, true, MT_)
#define MVX_CASES_MT(MT_) \
    MVX_CASES_CT(1, MT_)  \
    MVX_CASES_CT(4, MT_)  \
    MVX_CASES_CT(8, MT_)  \
    MVX_CASES_CT(16, MT_) \
    MVX_CASES_CT(32, MT_)
    MVX_CASES_MT(512)
    MVX_CASES_MT(1024)
#undef MVX_CASES_MT
#undef MVX_CASES_CT
#undef MVX_CASE
    return hipErrorInvalidValue;
}

// Dynamic LDS above the default 64 KB limit needs an opt-in per kernel and per device (only reached with more than
// 8 waves per workgroup, i.e. the MVX_NW experiment knob).
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
// bench: 5.9 us gaps before and after every voxelize launch). timed_launch() sets the pair, the next voxelize launch
// on this thread consumes it.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_launch_events(hipEvent_t start, hipEvent_t stop) {
    g_ev_start = start;
    g_ev_stop = stop;
}
bool launch_events_pending() { return g_ev_start != nullptr; }
template <typename K, typename... A>
static void launch_profiled(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    const hipEvent_t e0 = g_ev_start, e1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, e0, e1, 0u, args...);
    else hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
}

// blocks per molecule of the slab kernels
static unsigned slab_grid_x(const VoxParams &p) {
    const unsigned T = (unsigned)(p.nzc * p.nsy * p.nsx);
    if (p.xcd_ranges) return 8u * ((T + 7u) / 8u); // (mvx_slab_body.inc: every XCD a contiguous range of slabs)
    return T;
}

template <typename Ops, int MAXT = 1024, int WPE = 1>
static hipError_t launch_dense(const VoxArgs &a, size_t lds, unsigned grid, unsigned total, hipStream_t s) {
    static LdsLimit raised;
    const VoxParams &p = a.p;
    if (p.NW * 64 > MAXT) return hipErrorInvalidConfiguration;
    auto kern = &voxelize_dense_kernel<Ops, MAXT, WPE>;
    hipError_t e = raise_lds_limit(kern, lds, raised);
    if (e != hipSuccess) return e;
    launch_profiled(kern, dim3(grid), dim3(p.NW * 64), lds, s, a.rec, a.w, a.xlist, a.slist, a.slist_ext, a.offsets, a.n_one, a.Tc, a.kc,
                    a.out, a.p, (unsigned)(p.nzc * p.nsy * p.nsx), total);
    return hipGetLastError();
}

template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
constexpr bool mx_kernel() {
    return std::is_same<typename SlabOps<CT, GAUSS, CHANWISE, LANE_RANGE, false>::type, OpsMx32<GAUSS, LANE_RANGE, false>>::value;
}

struct GroupedFn {
    const VoxArgs &a;
    int32_t nb;
    hipStream_t s;
    template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        if constexpr (CT != 32 || CHANWISE) {
            return hipErrorInvalidValue;
        } else {
            VoxParams p = a.p;
            if (nb <= 0) return hipSuccess;
            if ((long long)nb * p.ncc > 65535) return hipErrorInvalidConfiguration;
            static LdsLimit raised;
            const size_t main_lds = voxelize_mx_lds_bytes(p.NW);
            p.dcap = (int32_t)main_lds; // where the slots' {T, k} table sits in LDS
            auto kern = &voxelize_kernel<CT, GAUSS, false, LANE_RANGE, MAXT, true>;
            hipError_t e = raise_lds_limit(kern, main_lds + 16 * CHAN_GROUP_SLOTS, raised);
            if (e != hipSuccess) return e;
            launch_profiled(kern, dim3(slab_grid_x(p), (unsigned)(nb * p.ncc)), dim3(p.NW * 64), main_lds + 16 * CHAN_GROUP_SLOTS, s, a.rec, a.w,
                            a.slist, a.slist_ext, a.Tc, a.kc, static_cast<float *>(a.out), p);
            return hipGetLastError();
        }
    }
};

struct LaunchFn {
    const VoxArgs &a;
    int32_t nb;
    hipStream_t s;
    template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        const VoxParams &p = a.p;
        if (nb <= 0) return hipSuccess;
        if ((long long)nb * p.ncc > 65535) return hipErrorInvalidConfiguration;
        static LdsLimit raised;
        constexpr bool mx = mx_kernel<CT, GAUSS, CHANWISE, LANE_RANGE>();
        const size_t lds = mx ? voxelize_mx_lds_bytes(p.NW) : voxelize_rounds_lds_bytes(CT, p.NW, MVX_CR);
        auto kern = &voxelize_kernel<CT, GAUSS, CHANWISE, LANE_RANGE, MAXT>;
        LdsLimit *state = &raised;
        if (!p.vec_store) {
            if constexpr (mx) {
                static LdsLimit raised_runs;
                kern = &voxelize_runs_kernel<GAUSS, MAXT>;
                state = &raised_runs;
            } else if (!LANE_RANGE) {
                return hipErrorInvalidConfiguration; // (only the per-lane-range kernels carry store_runs)
            }
        }
        hipError_t e = raise_lds_limit(kern, lds, *state);
        if (e != hipSuccess) return e;
#ifdef MVX_MOLMIX
        if (p.ncc == 1 && nb % MVX_MOLMIX == 0) {
            launch_profiled(kern, dim3((unsigned)(p.nzc * p.nsy * p.nsx * MVX_MOLMIX), (unsigned)(nb / MVX_MOLMIX)), dim3(p.NW * 64), lds, s,
                            a.rec, a.w, a.slist, a.slist_ext, a.Tc, a.kc, static_cast<float *>(a.out), a.p);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
#endif
        launch_profiled(kern, dim3(slab_grid_x(p), (unsigned)(nb * p.ncc)), dim3(p.NW * 64), lds, s, a.rec, a.w,
                        a.slist, a.slist_ext, a.Tc, a.kc, static_cast<float *>(a.out), a.p);
        return hipGetLastError();
    }
};

struct Dense64Fn {
    const VoxArgs &a;
    hipStream_t s;
    template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        const VoxParams &p = a.p;
        const long long total = (long long)p.B * p.ncc * p.nzc * p.nsy * p.nsx;
        if (total <= 0) return hipSuccess;
        if (total > 0xffffffffll) return hipErrorInvalidConfiguration;
        const unsigned grid = (unsigned)(total < 4096 ? total : 4096);
        if constexpr (CT > 16) {
            // 32 float64 accumulators per lane: 512-thread workgroups (the plan's slabs have at most 8 waves), so the
            // kernel may use 256 VGPRs; one chunk instead of two halves the staging, distance and exp work per slab
            if (p.NW > 8) return hipErrorInvalidValue;
            return launch_dense<OpsF64<CT, GAUSS, CHANWISE, LANE_RANGE>, 512>(a, dense64_lds_bytes(CT, p.NW), grid, (unsigned)total, s);
        } else {
            return launch_dense<OpsF64<CT, GAUSS, CHANWISE, LANE_RANGE>>(a, dense64_lds_bytes(CT, p.NW), grid, (unsigned)total, s);
        }
    }
};

struct DirectFn {
    const DirectArgs &d;
    const VoxParams &p;
    float *out;
    hipStream_t s;
    template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        if constexpr (MAXT != 512) {
            return hipErrorInvalidValue;
        } else {
            if (p.B <= 0) return hipSuccess;
            if ((long long)p.B * p.ncc > 65535) return hipErrorInvalidConfiguration;
            static LdsLimit raised;
            const size_t lds = direct_lds_bytes(CT, p.NW);
            auto kern = &voxelize_direct_kernel<CT, GAUSS, CHANWISE, LANE_RANGE, MAXT>;
            hipError_t e = raise_lds_limit(kern, lds, raised);
            if (e != hipSuccess) return e;
            launch_profiled(kern, dim3((unsigned)(p.nzc * p.nsy * p.nsx), (unsigned)(p.B * p.ncc)), dim3(p.NW * 64), lds, s, d, out, p);
            return hipGetLastError();
        }
    }
};

void scalar_radius_constants(double radius_scalar, float sigma32, bool gauss, double *T, float *k) {
    const float r32 = (float)radius_scalar;
    *T = d2_threshold(r32);
    *k = gauss && *T >= 0.0 ? gauss_coeff(r32, sigma32) : 0.0f;
}

hipError_t launch_voxelize_direct(const DirectArgs &d, const VoxParams &p, float *out, int32_t ct, bool gauss, bool chanwise,
                                  bool lane_range, hipStream_t s) {
    if (p.NW > 8) return hipErrorInvalidConfiguration;
    KernelKey k{ct, gauss, chanwise, chanwise ? true : lane_range, 512};
    return for_kernel(k, DirectFn{d, p, out, s});
}

hipError_t launch_voxelize(const VoxArgs &a, int32_t nb, int32_t ct, bool gauss, bool chanwise, bool lane_range, hipStream_t s) {
    KernelKey k{ct, gauss, chanwise, chanwise ? true : lane_range, a.p.NW <= 8 ? 512 : 1024};
    return for_kernel(k, LaunchFn{a, nb, s});
}

hipError_t launch_voxelize_grouped(const VoxArgs &a, int32_t nb, bool gauss, bool lane_range, hipStream_t s) {
    KernelKey k{32, gauss, false, lane_range, a.p.NW <= 8 ? 512 : 1024};
    return for_kernel(k, GroupedFn{a, nb, s});
}

template <bool GAUSS, bool LANE_RANGE>
static hipError_t launch_mx64(const VoxArgs &a, hipStream_t s) {
    static LdsLimit raised;
    VoxParams p = a.p;
    const size_t main_lds = voxelize_mx64_lds_bytes(p.NW), lds = main_lds + 512;
    p.dcap = (int32_t)main_lds; // where the kernel keeps its copy of the 2^(j/64) table
    auto kern = &voxelize64_kernel<GAUSS, LANE_RANGE, 512>;
    hipError_t e = raise_lds_limit(kern, lds, raised);
    if (e != hipSuccess) return e;
    const int per = 65535 / p.ncc; // molecules per launch (gridDim.y limit); the profiling bracket rides on the first launch
    for (int m0 = 0; m0 < p.B; m0 += per) {
        p.b0 = m0;
        const int nb = p.B - m0 < per ? p.B - m0 : per;
        launch_profiled(kern, dim3(slab_grid_x(p), (unsigned)(nb * p.ncc)), dim3(p.NW * 64), lds, s, a.rec, a.w, a.slist,
                        a.slist_ext, static_cast<double *>(a.out), p);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_voxelize64(const VoxArgs &a, int32_t ct, bool gauss, bool chanwise, bool lane_range, hipStream_t s) {
    // chunks of 32 channels with scalar / atom-wise radii on 8-wave slabs: the matrix-core slab kernel
    if (ct == 32 && !chanwise && a.p.NW <= 8 && a.p.dcap == 0) {
        if (gauss) return lane_range ? launch_mx64<true, true>(a, s) : launch_mx64<true, false>(a, s);
        return lane_range ? launch_mx64<false, true>(a, s) : launch_mx64<false, false>(a, s);
    }
    KernelKey k{ct, gauss, chanwise, chanwise ? true : lane_range, 512};
    return for_kernel(k, Dense64Fn{a, s});
}

} // namespace mvx
