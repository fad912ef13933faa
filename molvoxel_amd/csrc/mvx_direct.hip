// mvx_direct.hip - voxelize_direct_kernel: a whole per-molecule forward() call in ONE launch, one slab per workgroup (gfx950).
// Since round 4 only the per-lane-range instantiations are built: blockdims whose reference blocks cut through sub-tiles (4, 5,
// 12, ...). Every other per-molecule call takes
// voxelize_pair_kernel (mvx_pair.hip), which shares one leaner scan between two slabs.
#include "mvx_device.h"

namespace mvx {

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
//   B. rounds of up to 64 candidates: lane u < 8 of wave w prepares slot w + u*NW with prep_atom, as the prep kernel does
//      (sub-tiles that straddle reference blocks - blockdim 4, 5, 12, ... - need per-lane voxel ranges), and writes the
//      64-B record straight into the LDS row, while the wave's other lanes fetch the slot's channel weights from the
//      caller's feature rows (or build the one-hot / unit row of forward_types / forward_single); one barrier; every
//      wave then selects, one lane per row, the rows whose admitted z range reaches ITS sub-tile, and walks them
//      (Ops::accumulate, per-lane range checks); write-out as everywhere (Ops::write).
//   Any number of candidates and atoms works (rounds, segments); there is no overflow list and no dense kernel.
// LDS map: u16 list[NW*512] | int wcnt[16] | u32 pk[64] | union { scan strips ; rows ; tile }.
// (Channel-wise radii for features take the binned pipeline - the grouped matrix-core launch - whatever the call's size.)
constexpr int SEGW = 512;            // atoms one wave scans per segment
constexpr int SCAN_BLOCK = 128;      // atoms per coalesced fetch (3 x 1 KB)
constexpr int DIRECT_HDR_BYTES = 64 + 256; // wcnt + pk

static size_t direct_lds_bytes(int32_t ct, int32_t NW) {
    const size_t strips = (size_t)NW * SCAN_BLOCK * 12;
    const size_t un = voxelize_lds_bytes(ct, NW, DIRECT_CR);
    return (size_t)NW * SEGW * 2 + DIRECT_HDR_BYTES + (un > strips ? un : strips);
}

template <int CT, bool GAUSS, bool LANE_RANGE, int MAXT>
__global__ void __launch_bounds__(MAXT, DIRECT_WAVES_PER_SIMD)
    voxelize_direct_kernel(const DirectArgs A, float *__restrict__ out, const VoxParams P) {
    static_assert(LANE_RANGE, "only the per-lane-range instantiations are built (mvx_pair.hip serves uniform block culls)");
    typedef OpsF32<CT, GAUSS, LANE_RANGE> Ops;
    constexpr int SW = Ops::SW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = P.NW;
    unsigned short *list = reinterpret_cast<unsigned short *>(smem);
    int *wcnt = reinterpret_cast<int *>(smem + (size_t)NW * SEGW * 2);
    unsigned *pk = reinterpret_cast<unsigned *>(wcnt + 16);
    unsigned *un = pk + 64;
    float *strip = reinterpret_cast<float *>(un) + (size_t)wave * SCAN_BLOCK * 3; // this wave's scan strip

    const unsigned t = blockIdx.x;
    int b = (int)blockIdx.y, cc = 0;
    if (P.ncc > 1) {
        b = (int)__umulhi(blockIdx.y, P.ncc_inv); // blockIdx.y / ncc
        cc = (int)blockIdx.y - b * P.ncc;
    }
    int sx, sy, zc;
    decode_slab(t, P, sx, sy, zc);
    const int x0 = SUBX * sx, y0 = SUBY * sy, z0 = zc * SUBZ * NW;
    const int zt_lo = zc * NW, zt_hi = zt_lo + NW - 1;
    const int cbase = cc * CT; // first channel of this workgroup's chunk
    const PrepArgs &pa = A.pa;
    const int C = pa.C;

    int64_t a0 = 0, a1 = A.N;
    if (pa.offsets) {
        a0 = pa.offsets[b];
        a1 = pa.offsets[b + 1];
    }
    mvx_xform xf = pa.xf_one;
    if (pa.xforms) xf = pa.xforms[b];
    const bool has_xf = xf.flags != 0;

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
        if (pa.radii_src == RAD_SCALAR) region(std::false_type{});
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
                bool keep = prep_atom(pa, a, p, 0.0f, 0.0, R, rng);
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
            }
        }
        any = any || __ballot(kept) != 0ull; // (the same rows in every wave: workgroup-uniform)
        unsigned long long mask = __ballot(ok);
        while (mask) {
            const int sl = __builtin_ctzll(mask);
            mask &= mask - 1;
            Ops::accumulate(acc, un + sl * SW, L);
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
template <int CT, bool GAUSS>
static hipError_t launch_direct(const DirectArgs &d, const VoxParams &p, float *out, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if ((long long)p.B * p.ncc > 65535) return hipErrorInvalidConfiguration;
    static LdsLimit raised;
    const size_t lds = direct_lds_bytes(CT, p.NW);
    auto kern = &voxelize_direct_kernel<CT, GAUSS, true, 512>;
    hipError_t e = raise_lds_limit(kern, lds, raised);
    if (e != hipSuccess) return e;
    launch_profiled(kern, dim3((unsigned)(p.nzc * p.nsy * p.nsx), (unsigned)(p.B * p.ncc)), dim3(p.NW * 64), lds, s, d, out, p);
    return hipGetLastError();
}

// lane_range false - sub-tiles inside one reference block, the usual case - takes voxelize_pair_kernel (mvx_pair.hip), aligned
// grid or not; this kernel serves the per-lane-range cases (blockdim 4, 5, 12, ...).
hipError_t launch_voxelize_direct(const DirectArgs &d, const VoxParams &p, int64_t max_atoms, float *out, int32_t ct, bool gauss,
                                  bool lane_range, hipStream_t s) {
    if (p.NW > 8) return hipErrorInvalidConfiguration;
    if (!lane_range) return launch_voxelize_pair(d, p, max_atoms, out, ct, gauss, s);
#define MVX_CASE(CT_) \
    if (ct == CT_) return gauss ? launch_direct<CT_, true>(d, p, out, s) : launch_direct<CT_, false>(d, p, out, s);
    MVX_CASE(1)
    MVX_CASE(4)
    MVX_CASE(8)
    MVX_CASE(16)
    MVX_CASE(32)
#undef MVX_CASE
    return hipErrorInvalidValue;
}

void scalar_radius_constants(double radius_scalar, float sigma32, bool gauss, double *T, float *k) {
    const float r32 = (float)radius_scalar;
    *T = d2_threshold(r32);
    *k = gauss && *T >= 0.0 ? gauss_coeff(r32, sigma32) : 0.0f;
}

} // namespace mvx
