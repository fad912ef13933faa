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
//   prep_kernel      one thread per atom: rigid transform in the reference's fp64 op order, exact
//                    box cull + per-axis reference-block cull folded into an admitted voxel-index
//                    range, exact membership threshold T on d2, gaussian coefficient k.
//   voxelize_kernel  one workgroup per output slab of 4 x 4 x (4*NW) voxels (NW waves, one 4^3
//                    sub-tile per wave, one voxel per lane, CT channel accumulators per lane in
//                    registers). The workgroup scans the molecule's atom ranges for slab
//                    candidates (ordered, ballot/prefix compaction), stages candidate records +
//                    feature rows in LDS, every wave walks the candidates that touch its sub-tile
//                    (fp64 d2, compare with T, exp2, packed FMAs), then the accumulators are
//                    transposed through LDS and written with 16-B/lane stores in runs of full W
//                    rows. Every output byte is written exactly once, zeros included (the
//                    reference's overwrite semantics, numpy/voxelizer.py:133-135,158-160); no
//                    atomics, no memset, no (V, DHW) intermediate, no MFMA (scatter-reduce).
//
// Exactness: membership float32(float32(sqrt_f64(d2))/r32) <= 1 is equivalent to d2 <= T with
//   y  = largest fp64 whose float32 rounding is <= r32,  T = round_down(y * nextup(y))
// (derivation in DESIGN.md §4; checked against 20k radii on the CPU and by tests/test_hip_parity.py).
// d2 is formed exactly like scipy cdist: (dx*dx + dy*dy) + dz*dz in fp64 WITHOUT fma, so this TU
// must be compiled with -ffp-contract=off and without fast-math.
#include "mvx_internal.h"

#include <math.h>
#include <stdio.h>

// cdist-order arithmetic must not be fused, whatever flags the TU is built with.
#pragma clang fp contract(off)

namespace mvx {

typedef float float2v __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double next_up(double x) { // x > 0 finite
    return __longlong_as_double(__double_as_longlong(x) + 1);
}
__device__ __forceinline__ double next_down(double x) { // x > 0 finite
    return __longlong_as_double(__double_as_longlong(x) - 1);
}

// Largest fp64 d2 with float32(float32(sqrt(d2)) / r32) <= 1 (sqrt and division correctly rounded).
__device__ double d2_threshold(float r32) {
    if (!(r32 > 0.0f) || !(r32 < 3.0e38f)) return -1.0;
    const float up = __uint_as_float(__float_as_uint(r32) + 1u);
    const double m = 0.5 * ((double)r32 + (double)up); // midpoint between r32 and the next float (exact)
    const bool even = (__float_as_uint(r32) & 1u) == 0u;
    const double y = even ? m : next_down(m); // largest fp64 that rounds (ties-to-even) to <= r32
    const double yp = next_up(y);
    const double hi = y * yp;
    const double lo = fma(y, yp, -hi); // exact residual of the product
    return (lo >= 0.0) ? hi : next_down(hi);
}

__device__ __forceinline__ float gauss_coeff(float r32, float sigma32) {
    const double rs = (double)r32 * (double)sigma32;
    return (float)(-0.5 * 1.4426950408889634 / (rs * rs));
}

// do_transform in the reference's operation order (numpy/transform.py:44-60, _quaternion.py:24-50).
__device__ void apply_xform(const mvx_xform &xf, double &x, double &y, double &z) {
    if (xf.flags & MVX_XF_CENTER) {
        x = x - xf.center[0];
        y = y - xf.center[1];
        z = z - xf.center[2];
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
            x += xf.center[0];
            y += xf.center[1];
            z += xf.center[2];
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
__global__ void chan_aux_kernel(const float *radii, int C, int density, float sigma32, float *rmax, double *Tc,
                                float *kc) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float r = radii[c];
        Tc[c] = d2_threshold(r);
        kc[c] = density == MVX_GAUSSIAN ? gauss_coeff(r, sigma32) : 0.0f;
    }
    if (threadIdx.x == 0) {
        float m = radii[0];
        for (int c = 1; c < C; ++c) m = radii[c] > m ? radii[c] : m;
        rmax[0] = m;
    }
}

hipError_t launch_chan_aux(const float *radii, int32_t C, int32_t density, float sigma32, float *rmax, double *Tc,
                           float *kc, hipStream_t s) {
    hipLaunchKernelGGL(chan_aux_kernel, dim3(1), dim3(256), 0, s, radii, C, density, sigma32, rmax, Tc, kc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// prep: per-atom records
// ------------------------------------------------------------------------------------------------
// Admitted reference-block interval along one axis, as voxel indices (numpy/voxelizer.py:500-513):
// block b admits the atom iff (b == 0 or p > bounds[b-1] - r) and (b == nb-1 or p < bounds[b] + r),
// bounds[m] = axis[(m+1)*bd] + res/2 (numpy/voxelizer.py:55). Both conditions are monotone in b, so
// the admitted set is the interval [#(p >= bounds[m] + r), #(p > bounds[m] - r)].
__device__ __forceinline__ void block_interval(const Geom &g, double p, double r, int &vlo, int &vhi) {
    int bhi = 0, blo = 0;
    const double hres = g.res / 2.0;
    for (int m = 0; m < g.nb - 1; ++m) {
        const double ax = (double)((m + 1) * g.bd) * g.res - g.half;
        const double bound = ax + hres;
        if (p > bound - r) ++bhi;
        if (!(p < bound + r)) ++blo;
    }
    vlo = blo * g.bd;
    vhi = (bhi + 1) * g.bd - 1;
    if (vhi > g.D - 1) vhi = g.D - 1;
}

// lo = 0xffff, hi = 0: fails every overlap test (lo <= box_hi needs box_hi >= 65535, beyond any grid)
constexpr uint32_t EMPTY_RANGE = 0x0000ffffu;

__global__ void __launch_bounds__(256) prep_kernel(PrepArgs A) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= A.total) return;
    const int b = find_molecule(A.offsets, A.B, a);
    double p[3] = {A.coords[3 * a], A.coords[3 * a + 1], A.coords[3 * a + 2]};
    if (A.xforms) apply_xform(A.xforms[b], p[0], p[1], p[2]);

    const Geom g = A.g;
    const double ub = g.half, lb = -1 * g.half;
    float r32;   // membership radius (float32, as np.divide sees it)
    double rc;   // fp64 radius the culls use
    float rwin;  // widest radius for the conservative index window
    bool keep = true;
    int32_t type = 0;
    if (A.types) {
        type = A.types[a];
        if (type < 0 || type >= A.C) keep = false; // never index radii / channels out of range
    }
    if (A.radii_src == RAD_SCALAR) {
        rc = A.radius_scalar;
        r32 = (float)A.radius_scalar;
        rwin = r32;
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lb - rc) && (p[i] < ub + rc); // numpy/voxelizer.py:487-488
    } else if (A.radii_src == RAD_CHANNEL_FEATURES) {
        const float rmax = A.chan_aux[0];
        r32 = rmax;
        rwin = rmax;
        rc = (double)rmax;
        // np.float32 scalar: (python float -/+ float32) is evaluated in float32 (NEP 50), numpy/voxelizer.py:138
        const double lo = (double)((float)lb - rmax), hi = (double)((float)ub + rmax);
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] > lo) && (p[i] < hi);
    } else {
        r32 = (A.radii_src == RAD_ATOM) ? A.radii[a] : (keep ? A.radii[type] : 0.0f); // numpy/voxelizer.py:284-285
        rwin = r32;
        rc = (double)r32;
        for (int i = 0; i < 3; ++i) keep = keep && (p[i] + rc > lb) && (p[i] - rc < ub); // numpy/voxelizer.py:491-492
    }

    AtomRec R;
    R.px = p[0];
    R.py = p[1];
    R.pz = p[2];
    R.T = d2_threshold(r32);
    R.k = (A.density == MVX_GAUSSIAN) ? gauss_coeff(r32, A.sigma32) : 0.0f;
    R.type = type;
    R.pad[0] = R.pad[1] = R.pad[2] = 0;
    keep = keep && (R.T >= 0.0);

    uint32_t rng[3] = {EMPTY_RANGE, EMPTY_RANGE, EMPTY_RANGE};
    if (keep) {
        // Voxels that can pass |p - g_i| <= r are i in [ceil((p - r - g0)/res), floor((p + r - g0)/res)]; the
        // radius is widened by 1e-6 relative (fp64 rounding of this estimate is ~1e-15) so the window is a
        // superset of the membership set; membership itself is decided per voxel with the exact threshold.
        const double rr = (double)rwin * 1.000001 + 1e-9;
        for (int i = 0; i < 3; ++i) {
            double flo = ceil((p[i] - rr + g.half) / g.res);
            double fhi = floor((p[i] + rr + g.half) / g.res);
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
    A.rec[a] = R;
    A.bbox[a] = make_uint4(rng[0], rng[1], rng[2], 0u);
}

hipError_t launch_prep(const PrepArgs &a, hipStream_t s) {
    if (a.total <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((a.total + 255) / 256);
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
// x-slab binning: ordered list of the atoms whose admitted x-range touches each 4-voxel x-slab
// ------------------------------------------------------------------------------------------------
// One wave per (molecule, x-slab). Entries keep atom order (ballot + prefix compaction, no atomics,
// so downstream float sums are reproducible). Entry = {atom index in molecule, yr, zr, 0}; list
// (b, sx) lives at xlist[(b*nsx + sx) * xstride] (xstride = largest molecule of the batch), its length in
// xcount[b*nsx+sx].
__global__ void __launch_bounds__(64) xbin_kernel(const uint4 *bbox, const int64_t *offsets, int nsx, int xstride,
                                                  uint4 *xlist, int *xcount) {
    const int b = blockIdx.x / nsx, sx = blockIdx.x % nsx;
    const int lane = threadIdx.x;
    const int64_t a0 = offsets[b], a1 = offsets[b + 1];
    const int x0 = 4 * sx;
    uint4 *dst = xlist + (size_t)blockIdx.x * (size_t)xstride; // fixed stride: addressable from blockIdx alone
    int count = 0;
    for (int64_t base = a0; base < a1; base += 64) {
        const int64_t a = base + lane;
        bool m = false;
        uint4 bb = make_uint4(0, 0, 0, 0);
        if (a < a1) {
            bb = bbox[a];
            m = ((int)(bb.x & 0xffff) <= x0 + 3) && ((int)(bb.x >> 16) >= x0);
        }
        const unsigned long long mask = __ballot(m);
        if (m) {
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            dst[count + below] = make_uint4((unsigned)(a - a0), bb.y, bb.z, 0u);
        }
        count += __popcll(mask);
    }
    if (lane == 0) xcount[blockIdx.x] = count;
}

hipError_t launch_xbin(const uint4 *bbox, const int64_t *offsets, int32_t B, int32_t nsx, int32_t xstride, uint4 *xlist,
                       int *xcount, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(xbin_kernel, dim3((unsigned)(B * nsx)), dim3(64), 0, s, bbox, offsets, nsx, xstride, xlist,
                       xcount);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// voxelize
// ------------------------------------------------------------------------------------------------
// One workgroup = one slab of 4 x 4 x (4*NW) voxels; one wave = one 4^3 sub-tile; one lane = one voxel with
// CT channel accumulators in registers.
//   1. scan: every thread loads one entry of the slab's x-list (address known from blockIdx alone, so the load
//      is in flight while the list length is still being fetched) and tests its y/z ranges against the slab;
//      matches are compacted in atom order (ballot + cross-wave prefix, one barrier) into an LDS list;
//   2. stage: wave w copies candidates w, w+NW, ... (64-B record + CT feature floats, one coalesced load each,
//      up to 8 in flight) into LDS; one barrier;
//   3. walk: each wave picks the candidates whose z range touches its sub-tile (lane-parallel filter + ballot)
//      and processes them: broadcast LDS reads, fp64 d2 in cdist order, compare with T, exp2, packed FMAs;
//   4. write-out: accumulators -> LDS tile (CR = min(CT,16) channels per round) -> 16-B/lane stores in
//      whole-row runs. Empty slabs skip the LDS round trip.
// LDS map (dynamic, 16-B aligned), LCAP = 64 * NW:
//   [0, 4*LCAP) int list[] | [4*LCAP, 8*LCAP) uint32 zr[] | [8*LCAP, +128) int wcnt[2][16] |
//   union { dcap x STRIDE candidate bytes ; (CR*16 rows) x RS floats out tile, RS = 4*NW + pad ((RS/4) odd) }
__host__ __device__ __forceinline__ int row_stride_floats(int NW) { return 4 * NW + ((NW & 1) ? 8 : 4); }
__host__ __device__ __forceinline__ int cand_stride_bytes(int ct) { return 64 + (ct < 4 ? 16 : 4 * ct); }

size_t voxelize_lds_bytes(int32_t ct, int32_t NW, int32_t *dcap) {
    const int cr = ct < 16 ? ct : 16;
    const int stride = cand_stride_bytes(ct);
    size_t un = (size_t)cr * 16 * row_stride_floats(NW) * 4;
    const size_t min_cand = (size_t)96 * stride;
    if (un < min_cand) un = min_cand;
    int cap = (int)(un / stride);
    if (cap > 64 * NW) cap = 64 * NW;
    if (dcap) *dcap = cap;
    return (size_t)8 * 64 * NW + 128 + un;
}

// VARIANT: 0 = every channel chunk is full (C % CT == 0) and one radius per atom: fast path;
//          1 = partial channel chunks (any C); 2 = channel-wise radii (per-channel membership, any C)
template <int CT, int MODE, bool GAUSS, int VARIANT, bool LANE_RANGE, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 6 : 4))
    voxelize_kernel(const AtomRec *__restrict__ rec, const float *__restrict__ features,
                    const uint4 *__restrict__ xlist, const int *__restrict__ xcount,
                    const int64_t *__restrict__ offsets, const double *__restrict__ Tc,
                    const float *__restrict__ kc, float *__restrict__ out, const VoxParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CR = CT < 16 ? CT : 16; // channels per write-out round
    constexpr int NROUND = CT / CR;
    constexpr int STRIDE = 64 + (CT < 4 ? 16 : 4 * CT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = P.NW;
    const int nthreads = NW * 64;
    const int LCAP = nthreads;
    const int D = P.D;

    int *list = reinterpret_cast<int *>(smem);
    unsigned *zr_l = reinterpret_cast<unsigned *>(smem + 4 * LCAP);
    int *wcnt = reinterpret_cast<int *>(smem + 8 * LCAP);
    char *un = smem + 8 * LCAP + 128;
    float *tile = reinterpret_cast<float *>(un);

    // ---- block -> (molecule, channel chunk, slab) -------------------------------------------------
    // Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only). Remap so
    // that the nzc z-chunks of one (x, y) column run on the same XCD back to back: their runs are pieces of the
    // same rows, so that XCD's L2 sees whole rows. Bijective on the leading multiple of 8*nzc.
    int bid = blockIdx.x;
    if (P.xcd_remap) {
        const int G = 8 * P.nzc;
        const int full = (int)(gridDim.x / G) * G;
        if (bid < full) {
            const int xcd = bid & 7, slot = bid >> 3;
            bid = ((slot / P.nzc) * 8 + xcd) * P.nzc + (slot % P.nzc);
        }
    }
    const int zc = bid % P.nzc;
    bid /= P.nzc;
    const int sy = bid % P.nsx;
    bid /= P.nsx;
    const int sx = bid % P.nsx;
    bid /= P.nsx;
    const int cc = bid % P.ncc;
    const int b = bid / P.ncc;
    const int cbase = cc * 32;

    const int x0 = 4 * sx, y0 = 4 * sy, z0 = zc * 4 * NW;
    const int zhi_slab = z0 + 4 * NW - 1;
    // x-list of (b, sx): fixed-stride region, so the first entry load does not wait for any other load
    const uint4 *__restrict__ xl = xlist + ((size_t)b * P.nsx + sx) * (size_t)P.xstride;
    uint4 e0 = make_uint4(0u, 0x0000ffffu, 0x0000ffffu, 0u);
    if (tid < P.xstride) e0 = xl[tid];
    const int nx = (P.ablate & 2) ? 0 : xcount[b * P.nsx + sx];
    const int64_t a0 = offsets[b];

    // ---- this lane's voxel ---------------------------------------------------------------------
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int ix = x0 + lx, iy = y0 + ly, iz = z0 + 4 * wave + lz;
    const double gx = (double)ix * P.res - P.half; // axis[i] = i*res - width/2, numpy/voxelizer.py:41-43
    const double gy = (double)iy * P.res - P.half;
    const double gz = (double)iz * P.res - P.half;
    const int zlo_w = z0 + 4 * wave, zhi_w = zlo_w + 3;

    float2v acc[(CT + 1) / 2];
#pragma unroll
    for (int c = 0; c < (CT + 1) / 2; ++c) acc[c] = (float2v){0.0f, 0.0f};

    bool any_candidate = false;
    int cursor = 0;
    int phase = 0;
    while (cursor < nx) {
        // ---- 1. ordered compaction of the x-slab list against this slab's y/z box ----------------
        int nlist = 0;
        while (cursor < nx && nlist + nthreads <= LCAP) {
            const int i = cursor + tid;
            bool m = false;
            uint4 e = e0;
            if (cursor > 0 && i < nx) e = xl[i];
            if (i < nx)
                m = ((int)(e.y & 0xffff) <= y0 + 3) && ((int)(e.y >> 16) >= y0) && ((int)(e.z & 0xffff) <= zhi_slab) &&
                    ((int)(e.z >> 16) >= z0);
            const unsigned long long mask = __ballot(m);
            int *wc = wcnt + (phase & 1) * 16;
            if (lane == 0) wc[wave] = __popcll(mask);
            __syncthreads();
            int pre = 0, tot = 0;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                const int4 c4 = *reinterpret_cast<const int4 *>(wc + 4 * w4);
                const int cs[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int w = 4 * w4 + k;
                    const int c = (w < NW) ? cs[k] : 0;
                    tot += c;
                    pre += (w < wave) ? c : 0;
                }
            }
            if (m) {
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                const int pos = nlist + pre + below;
                list[pos] = (int)e.x;
                zr_l[pos] = e.z;
            }
            nlist += tot;
            cursor += nthreads;
            ++phase;
        }
        __syncthreads(); // list complete
        if (nlist == 0) continue;
        any_candidate = true;

        for (int c0 = 0; c0 < nlist; c0 += P.dcap) {
            const int n = (nlist - c0) < P.dcap ? (nlist - c0) : P.dcap;
            // ---- 2. stage: lanes 0-15 copy the 64-B record, lanes 16..16+CT-1 the feature row ---
            for (int j0 = wave; j0 < n; j0 += 8 * NW) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * NW;
                    v[u] = 0u;
                    if (j < n) {
                        const int64_t a = a0 + list[c0 + j];
                        if (lane < 16) {
                            v[u] = reinterpret_cast<const unsigned *>(rec + a)[lane];
                        } else if (lane < 16 + CT) {
                            const int c = lane - 16;
                            float f = 0.0f;
                            if (cbase + c < P.C) {
                                if (MODE == MODE_FEATURES) f = features[a * P.C + cbase + c];
                                else if (MODE == MODE_TYPES) f = (rec[a].type == cbase + c) ? 1.0f : 0.0f;
                                else f = 1.0f;
                            }
                            v[u] = __float_as_uint(f);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * NW;
                    if (j < n && lane < 16 + CT) reinterpret_cast<unsigned *>(un + (size_t)j * STRIDE)[lane] = v[u];
                }
            }
            __syncthreads();

            // ---- 3. walk the candidates that touch this wave's sub-tile ------------------------------
            for (int jb = 0; jb < ((P.ablate & 1) ? 0 : n); jb += 64) {
                const int j = jb + lane;
                bool ok = false;
                if (j < n) {
                    const unsigned zr = zr_l[c0 + j];
                    ok = ((int)(zr & 0xffff) <= zhi_w) && ((int)(zr >> 16) >= zlo_w);
                }
                unsigned long long mask = __ballot(ok);
                while (mask) {
                    const int jj = jb + __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const char *r = un + (size_t)jj * STRIDE;
                    const double2 Pxy = *reinterpret_cast<const double2 *>(r);      // px, py
                    const double2 PzT = *reinterpret_cast<const double2 *>(r + 16); // pz, T
                    const uint4 q = *reinterpret_cast<const uint4 *>(r + 32);       // k, type, xr, yr
                    const double dx = Pxy.x - gx, dy = Pxy.y - gy, dz = PzT.x - gz;
                    const double d2 = (dx * dx + dy * dy) + dz * dz; // cdist order, no fma
                    bool hit = d2 <= PzT.y;
                    if (LANE_RANGE) {
                        const unsigned zr = *reinterpret_cast<const unsigned *>(r + 48);
                        hit = hit && (ix >= (int)(q.z & 0xffff)) && (ix <= (int)(q.z >> 16)) &&
                              (iy >= (int)(q.w & 0xffff)) && (iy <= (int)(q.w >> 16)) && (iz >= (int)(zr & 0xffff)) &&
                              (iz <= (int)(zr >> 16));
                    }
                    const float d2f = (float)d2;
                    const float *f = reinterpret_cast<const float *>(r + 64);
                    float val = 0.0f;
                    if (VARIANT != 2) {
                        const float e = GAUSS ? __builtin_amdgcn_exp2f(__uint_as_float(q.x) * d2f) : 1.0f;
                        val = hit ? e : 0.0f;
                    }
                    if constexpr (CT == 1) {
                        float vc = val;
                        if (VARIANT == 2) {
                            const float e = GAUSS ? __builtin_amdgcn_exp2f(kc[cbase] * d2f) : 1.0f;
                            vc = (hit && d2 <= Tc[cbase]) ? e : 0.0f;
                        }
                        acc[0].x = fmaf(vc, f[0], acc[0].x);
                    } else if constexpr (VARIANT != 2) {
                        const float2v v2 = (float2v){val, val};
#pragma unroll
                        for (int c = 0; c < CT / 2; ++c) {
                            const float2v f2 = *reinterpret_cast<const float2v *>(f + 2 * c);
                            acc[c] = __builtin_elementwise_fma(v2, f2, acc[c]);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            const int ch = (cbase + c < P.C) ? cbase + c : P.C - 1;
                            const float e = GAUSS ? __builtin_amdgcn_exp2f(kc[ch] * d2f) : 1.0f;
                            const float vc = (hit && d2 <= Tc[ch]) ? e : 0.0f;
                            if (c & 1) acc[c / 2].y = fmaf(vc, f[c], acc[c / 2].y);
                            else acc[c / 2].x = fmaf(vc, f[c], acc[c / 2].x);
                        }
                    }
                }
            }
            __syncthreads(); // candidates consumed: the union region / list may be rewritten
        }
    }

    // ---- 4. write-out ----------------------------------------------------------------------------
    const int RS = row_stride_floats(NW);
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    const int q = tid % NW;      // float4 slot inside a row
    const int rfirst = tid / NW; // 0..63: row of this thread in pass 0; rows advance by 64 (= 4 channels) per pass
    const int zq = z0 + 4 * q;
    const int sxx = (rfirst >> 2) & 3, syy = rfirst & 3, cfirst = rfirst >> 4;
    const bool vox_ok = (x0 + sxx < D) && (y0 + syy < D) && (zq < D);
    float *dst0 = out + ((size_t)b * P.C + cbase + cfirst) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
    if (!any_candidate) {
        // empty slab: pure zero fill with the same addressing (no LDS round trip)
        if (vox_ok) {
#pragma unroll 4
            for (int c = cfirst; c < CT; c += 4) {
                if (cbase + c >= P.C) break;
                float *dst = dst0 + (size_t)(c - cfirst) * D3;
                if (P.vec_store) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (zq + e < D) dst[e] = 0.0f;
                }
            }
        }
        return;
    }
    const int col = 4 * wave + lz;
    const int rxy = lx * 4 + ly;
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        if (rd > 0) __syncthreads(); // previous round fully read
#pragma unroll
        for (int c = 0; c < CR; ++c) {
            const int cg = rd * CR + c;
            const float v = (cg & 1) ? acc[cg / 2].y : acc[cg / 2].x;
            tile[(c * 16 + rxy) * RS + col] = v;
        }
        __syncthreads();
        if (vox_ok) {
#pragma unroll
            for (int p = 0; p < (CR + 3) / 4; ++p) {
                const int c = cfirst + 4 * p; // channel inside the round
                if (c >= CR || cbase + rd * CR + c >= P.C) break;
                const float4 v = *reinterpret_cast<const float4 *>(tile + (rfirst + 64 * p) * RS + 4 * q);
                float *dst = dst0 + (size_t)(rd * CR + 4 * p) * D3;
                if (P.vec_store) {
                    *reinterpret_cast<float4 *>(dst) = v;
                } else {
                    const float e4[4] = {v.x, v.y, v.z, v.w};
                    for (int e = 0; e < 4; ++e)
                        if (zq + e < D) dst[e] = e4[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
struct KernelKey {
    int ct, mode;
    bool gauss;
    int variant;
    bool lane_range;
    int maxt;
};

// Calls fn.template operator()<CT, MODE, GAUSS, VARIANT, LANE_RANGE, MAXT>() for the instantiation `k` names.
template <typename Fn>
static hipError_t for_kernel(const KernelKey &k, Fn &&fn) {
#define MVX_CASE(CT_, MODE_, G_, V_, LR_, MT_)                                                               \
    if (k.ct == CT_ && k.mode == MODE_ && k.gauss == G_ && k.variant == V_ && k.lane_range == LR_ && k.maxt == MT_) \
        return fn.template operator()<CT_, MODE_, G_, V_, LR_, MT_>();
#define MVX_CASES_GL(CT_, MODE_, V_, MT_)     \
    MVX_CASE(CT_, MODE_, true, V_, false, MT_)  \
    MVX_CASE(CT_, MODE_, false, V_, false, MT_) \
    MVX_CASE(CT_, MODE_, true, V_, true, MT_)   \
    MVX_CASE(CT_, MODE_, false, V_, true, MT_)
#define MVX_CASES_CT(CT_, MT_)                      \
    MVX_CASES_GL(CT_, MODE_FEATURES, 0, MT_)        \
    MVX_CASES_GL(CT_, MODE_FEATURES, 1, MT_)        \
    MVX_CASES_GL(CT_, MODE_TYPES, 0, MT_)           \
    MVX_CASE(CT_, MODE_FEATURES, true, 2, true, MT_)  \
    MVX_CASE(CT_, MODE_FEATURES, false, 2, true, MT_)
#define MVX_CASES_MT(MT_)    \
    MVX_CASES_CT(1, MT_)     \
    MVX_CASES_CT(4, MT_)     \
    MVX_CASES_CT(8, MT_)     \
    MVX_CASES_CT(16, MT_)    \
    MVX_CASES_CT(32, MT_)    \
    MVX_CASES_GL(1, MODE_SINGLE, 0, MT_)
    MVX_CASES_MT(512)
    MVX_CASES_MT(1024)
#undef MVX_CASES_MT
#undef MVX_CASES_CT
#undef MVX_CASES_GL
#undef MVX_CASE
    return hipErrorInvalidValue;
}

struct LaunchFn {
    const VoxArgs &a;
    hipStream_t s;
    template <int CT, int MODE, bool GAUSS, int VARIANT, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        const long long blocks = (long long)a.p.B * a.p.ncc * a.p.nsx * a.p.nsx * a.p.nzc;
        if (blocks <= 0) return hipSuccess;
        if (blocks > 0x7fffffffLL) return hipErrorInvalidConfiguration;
        const size_t lds = voxelize_lds_bytes(CT, a.p.NW, nullptr);
        if (lds > 64 * 1024) { // above the default dynamic-LDS limit: raise it once per instantiation
            static size_t raised = 0;
            if (lds > raised) {
                hipError_t e = hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&voxelize_kernel<CT, MODE, GAUSS, VARIANT, LANE_RANGE, MAXT>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                raised = lds;
            }
        }
        hipLaunchKernelGGL((voxelize_kernel<CT, MODE, GAUSS, VARIANT, LANE_RANGE, MAXT>), dim3((unsigned)blocks),
                           dim3(a.p.NW * 64), lds, s, a.rec, a.features, a.xlist, a.xcount, a.offsets, a.Tc, a.kc,
                           a.out, a.p);
        return hipGetLastError();
    }
};

hipError_t launch_voxelize(const VoxArgs &a, int32_t ct, bool gauss, bool chanwise, bool lane_range, hipStream_t s) {
    int variant = 0;
    if (chanwise) variant = 2;
    else if (a.p.mode == MODE_FEATURES && a.p.C % ct != 0) variant = 1;
    KernelKey k{ct, a.p.mode, gauss, variant, variant == 2 ? true : lane_range, a.p.NW <= 8 ? 512 : 1024};
    return for_kernel(k, LaunchFn{a, s});
}

hipError_t configure_kernels() { return hipSuccess; }

} // namespace mvx
