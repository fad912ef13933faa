// mvx_slab.hip - the batched float32 voxelize kernels (gfx950): one workgroup per output slab.
//
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
//                    the vector ALU (OpsF32, packed FMAs).
//   voxelize_runs_kernel  the same walk for grids whose rows are not whole 16-byte quads (run-wise write-out).
#include "mvx_device.h"
#include "mvx_ops32.h"

#include <algorithm>

namespace mvx {

#ifdef MVX_DIAG
hipError_t set_diag_buffer(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &p, sizeof(p)); }
#endif

size_t voxelize_lds_bytes(int32_t ct, int32_t NW, int32_t crmax) {
    const int cr = ct < crmax ? ct : crmax;
    const size_t tile = (size_t)cr * RPC * row_stride_floats(NW) * 4;
    const size_t cand = (size_t)64 * cand_stride_words(ct) * 4;
    return tile > cand ? tile : cand;
}

// (the matrix-core kernels keep the rows of TWO rounds: stage_first_rounds)
static size_t voxelize_mx_lds_bytes(int32_t NW) { return std::max(voxelize_lds_bytes(32, NW, MX_CR), (size_t)2 * 64 * cand_stride_words(32) * 4); }

// voxelize_kernel's arithmetic: 32-channel chunks go to the matrix cores (OpsMx32), narrower chunks to the vector ALU in
// candidate pairs (OpsPair); the per-lane-range variants keep one voxel per lane and candidate (OpsF32)
template <int CT, bool GAUSS, bool LANE_RANGE, bool GROUPED>
struct SlabOps {
    typedef OpsF32<CT, GAUSS, LANE_RANGE> type;
};
template <int CT, bool GAUSS>
struct SlabOps<CT, GAUSS, false, false> {
    typedef OpsPair<CT, GAUSS> type;
};
// (not the per-lane-range variants - blockdim 4, 5, 12, ...: their six extra index comparisons per voxel do not fit the
// 64 registers of the two-voxel layout without scratch: 0.527 against 0.479 ms per 64 cfg-2 molecules at blockdim 5)
template <bool GAUSS>
struct SlabOps<32, GAUSS, false, false> {
    typedef OpsMx32<GAUSS, false, false> type;
};
template <bool GAUSS>
struct SlabOps<32, GAUSS, false, true> {
    typedef OpsMx32<GAUSS, false, true> type;
};
// (grouped launches - channel-wise features by radius - exist on the matrix-core path only, per-lane ranges or not)
template <bool GAUSS>
struct SlabOps<32, GAUSS, true, true> {
    typedef OpsMx32<GAUSS, true, true> type;
};

template <int CT, bool GAUSS, bool LANE_RANGE, int MAXT, bool GROUPED = false>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 8 : BIG_WAVES_PER_SIMD))
    voxelize_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                    const uint2 *__restrict__ slist_ext, const double *__restrict__ Tc, const float *__restrict__ kc, float *__restrict__ out,
                    const VoxParams P) {
    typedef typename SlabOps<CT, GAUSS, LANE_RANGE, GROUPED>::type Ops;
#include "mvx_slab_body.inc"
}

// ---- voxelize_narrow_kernel: narrow chunks with NSUB (2 or 4) sub-tiles per wave -------------------------------------------
// What is left of a narrow launch after OpsPair is per-wave fixed cost: at cfg-3 density a wave walks two or three candidate
// pairs (~85 vector instructions) around ~110 of prologue, staging, row filter and write-out, and as many scalar ones. Here a
// slab of NS sub-tiles is served by NS / NSUB waves: wave w owns sub-tiles NSUB w ... NSUB w + NSUB - 1 - one prologue, one
// share of the staging (rows two per load, vector-ALU addresses, as stage_round_v), then filter + pair walk once per
// sub-tile (OpsPair::walk with that sub-tile's z coordinate) into NSUB accumulator sets, and one write-out. Same candidates,
// same order, same arithmetic per voxel as voxelize_kernel<CT < 32>: bit-identical grids. At most 8 waves per workgroup
// whatever the row length (16 sub-tiles: 128 voxels), so there is no 1024-thread variant. Same box, one -> two sub-tiles
// per wave, kernel: forward_single 0.118 -> 0.109 ms, 8 types 0.173 -> 0.162, cfg-3 x 256 0.219 -> 0.164 (profiles/r04_narrow.txt).
// Waves per SIMD the kernel is compiled for: its natural register need with two sub-tiles is 66 / 72 / 80 for 1 / 4 / 8
// channels (accumulator sets, eight row loads in flight); at the 64 of voxelize_kernel it spilled 1 ... 17 registers.
// (16 channels: 96 registers and no gain over one sub-tile per wave, 0.228 ms both - they keep voxelize_kernel.)
constexpr int narrow_waves_per_simd(int ct, int nsub) { return 8; }
template <int CT, bool GAUSS, int NSUB>
__global__ void __launch_bounds__(512, narrow_waves_per_simd(CT, NSUB))
    voxelize_narrow_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                           const uint2 *__restrict__ slist_ext, float *__restrict__ out, const VoxParams P) {
    typedef OpsPair<CT, GAUSS> Ops;
    constexpr int SW = Ops::SW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *un = reinterpret_cast<unsigned *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = P.NW, NWV = NS / NSUB; // sub-tiles per slab, waves per workgroup

    unsigned t = blockIdx.x;
    if (P.nzc > 1) { // consecutive blocks = consecutive x-slabs (mvx_slab_body.inc)
        const unsigned per_x = (unsigned)(P.nsy * P.nzc), nsx = (unsigned)P.nsx;
        const unsigned tq = __umulhi(t, P.nsx_inv); // t / nsx (nsx > 1 here: several slabs per row means D > 64)
        t = (t - tq * nsx) * per_x + tq;
    }
    int b = (int)blockIdx.y, cc = 0;
    if (P.ncc > 1) {
        b = (int)__umulhi(blockIdx.y, P.ncc_inv); // blockIdx.y / ncc
        cc = (int)blockIdx.y - b * P.ncc;
    }
    b += P.b0;
    const uint2 *__restrict__ line = slist + ((size_t)b * (size_t)P.nslab + t) * SLOTS; // (uniform: scalar loads)
    const uint2 hdr = line[0];
    int sx, sy, zc;
    decode_slab(t, P, sx, sy, zc);
    const int x0 = SUBX * sx, y0 = SUBY * sy, z0 = zc * SUBZ * NS;
    const int cbase = P.c0 + cc * CT;
    // voxel centres: both sub-tiles share x and y and differ in z only
    LaneCtx L0 = Ops::ctx(lane, NSUB * wave, x0, y0, z0, zc * NS, cbase, P);
    double gz[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) gz[s] = (double)(L0.iz + SUBZ * s) * P.res - P.half;
    typename Ops::Acc acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) Ops::zero(acc[s]);

    const unsigned n_hdr = __builtin_amdgcn_readfirstlane(hdr.x);
    const unsigned first = __builtin_amdgcn_readfirstlane(hdr.y); // the molecule's first atom (atom indices fit 31 bits)
    if (n_hdr > 0) {
        const bool xl = __builtin_expect(n_hdr > (unsigned)LINE_CAP, 0); // the slab walks its (molecule, x-slab) list (mvx_slab_body.inc)
        const uint2 *__restrict__ ext = slist_ext + ((size_t)b * (size_t)P.nslab + t) * EXT_SLOTS;
        int n = (int)n_hdr;
        if (xl) {
            const uint2 where = line[1];
            const unsigned long long at = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)where.y) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)where.x);
            typedef const uint2 __attribute__((address_space(1))) *global_u2;
            const global_u2 xlp = (global_u2)at;
            n = __builtin_amdgcn_readfirstlane((int)xlp[0].x);
            line = (const uint2 *)(xlp + (XL_HEADER - 1));
            ext = line + SLOTS;
        }
        const int half = lane >> 5, wd = lane & 31;
        const bool used = wd < 16 + Ops::WW;
        const unsigned *base = wd < 16 ? rec + wd : w + (cbase + wd - 16);
        const unsigned stride = wd < 16 ? 16u : (unsigned)P.w_stride;
        const int RW = 8 * NS < 64 ? 8 * NS : 64; // rows staged per round
        for (int e0 = 0; e0 <= n; e0 += RW) {
            if (e0 > 0) __syncthreads(); // every wave is done with the previous round's rows
            {   // stage: the round's entries one per lane, rows two per load instruction. Branch-free inside: a slot without a
                // candidate fetches the molecule's first row instead and lands in a dump row behind the round's rows (sixteen
                // exec-mask branches per wave before: 142 scalar instructions in this block for the D = 48 kernel)
                const int e = e0 + lane;
                int ai = 0;
                if (lane < RW && e >= 1 && e <= n) ai = (int)(e < SLOTS ? line[e].x : ext[e - SLOTS].x);
                const int lo = (1 - e0) > 0 ? (1 - e0) : 0, hi = (n - e0) < RW - 1 ? (n - e0) : RW - 1; // slot sl holds a candidate iff lo <= sl <= hi
                const unsigned span = (unsigned)(hi - lo);
                const int dsl = 2 * NWV;
                for (int sl0 = wave + NWV * half; sl0 - NWV * half - wave < RW; sl0 += 8 * dsl) { // (one trip for 4 waves: 8 loads in flight)
                    unsigned a[8], v[8];
                    int slot[8];
                    int sl = sl0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { // (every lane takes part: ds_bpermute reads nothing from a masked-off source lane)
                        const unsigned ar = (unsigned)__builtin_amdgcn_ds_bpermute(4 * sl, ai);
                        const bool in = (unsigned)(sl - lo) <= span;
                        a[i] = in ? ar : 0u;
                        slot[i] = in ? sl : RW;
                        sl += dsl;
                        asm volatile("" : "+v"(sl)); // (one add per slot: left alone the compiler rebuilds wave + NWV * (half + 2 i) with a quarter-rate multiply each)
                    }
                    if (used) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = base[(size_t)(first + a[i]) * stride];
#pragma unroll
                        for (int i = 0; i < 8; ++i) un[slot[i] * SW + wd] = v[i];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NSUB; ++s) { // one lane per staged row decides whether this sub-tile takes it; then the pair walk
                bool ok = false;
                if (lane < RW && e0 + lane >= 1 && e0 + lane <= n) {
                    const unsigned *r = un + lane * SW;
                    const unsigned zr = r[12];
                    ok = ((int)((zr & 0xffff) >> SUBZ_SH) <= L0.zt_w + s) && ((int)((zr >> 16) >> SUBZ_SH) >= L0.zt_w + s);
                    if (xl) {
                        const unsigned yr = r[11];
                        ok = ok && ((int)((yr & 0xffff) >> SUBY_SH) <= sy) && ((int)((yr >> 16) >> SUBY_SH) >= sy);
                    }
                }
                LaneCtx Ls = L0;
                Ls.gz = gz[s];
                Ops::walk(acc[s], __ballot(ok), un, lane, Ls, P, nullptr, nullptr);
            }
        }
    }

    // ---- write-out: [channel][x, y row][z] tile, CR_F32 channels per round, read back as float4 rows of the whole slab ----
    constexpr int CR = CT < CR_F32 ? CT : CR_F32, NROUND = CT / CR;
    const int D = P.D, RS = row_stride_floats(NS);
    const size_t D2 = (size_t)D * D, D3 = D2 * D;
    const int F4 = (SUBZ / 4) * NS;          // float4 slots per row
    // this thread's slot and first row; rows advance by nthr / F4 = 32 / NSUB per pass. (tid / F4 without the ~40-instruction
    // integer division: (tid + 0.5) / F4 lies at least 1 / 64 away from every integer, far beyond the float rounding)
    const int rfirst = (int)(((float)tid + 0.5f) * __frcp_rn((float)F4));
    const int q = tid - rfirst * F4;
    constexpr int CPP = 32 / NSUB / RPC;       // ... i.e. by CPP = 2 | 1 channels
    static_assert(NSUB == 2 || NSUB == 4, "read-back passes of whole channels");
    const int zq = z0 + 4 * q;
    const int sxx = (rfirst >> SUBY_SH) & (SUBX - 1), syy = rfirst & (SUBY - 1), cfirst = rfirst / RPC; // cfirst < CPP
    const bool vox_ok = (x0 + sxx < D) && (y0 + syy < D) && (zq < D);
    float *dst0 = out + ((size_t)b * P.C + cbase + cfirst) * D3 + (size_t)(x0 + sxx) * D2 + (size_t)(y0 + syy) * D + zq;
    if (n_hdr == 0) { // zero fill without the LDS round trip, held back and sent in pieces in big launches (write_slab)
        if (P.pace) __builtin_amdgcn_s_sleep(EMPTY_HOLD);
        if (vox_ok) {
#pragma unroll
            for (int p = 0; p < (CT + CPP - 1) / CPP; ++p) {
                const int c = cfirst + CPP * p;
                if (c < CT && cbase + c < P.C) store_f4(dst0 + (size_t)(CPP * p) * D3, make_float4(0.f, 0.f, 0.f, 0.f));
                if (P.pace && ((p + 1) * CPP) % 8 == 0 && p + 1 < (CT + CPP - 1) / CPP) __builtin_amdgcn_s_sleep(EMPTY_SPLIT);
            }
        }
        return;
    }
    float *tile = reinterpret_cast<float *>(un);
    const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1), lx = lane >> (SUBZ_SH + SUBY_SH);
    const int rxy = lx * SUBY + ly;
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        __syncthreads(); // candidate rows (round 0) / previous tile (later rounds) fully consumed
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int col = SUBZ * (NSUB * wave + s) + lz;
#pragma unroll
            for (int c = 0; c < CR; ++c) {
                const int cg = rd * CR + c;
                tile[(c * RPC + rxy) * RS + col] = (cg & 1) ? acc[s][cg / 2].y : acc[s][cg / 2].x;
            }
        }
        __syncthreads();
        if (vox_ok) {
#pragma unroll
            for (int p = 0; p < (CR + CPP - 1) / CPP; ++p) {
                const int c = cfirst + CPP * p; // channel inside the round
                if (c < CR && cbase + rd * CR + c < P.C) {
                    const float4 v = *reinterpret_cast<const float4 *>(tile + (rfirst + CPP * RPC * p) * RS + 4 * q);
                    store_f4(dst0 + (size_t)(rd * CR + CPP * p) * D3, v);
                }
            }
        }
    }
}

// 32-channel chunks of float32 grids whose rows are not whole 16-byte quads (odd dimensions, unaligned grid slices): the
// matrix-core walk with the run-wise write-out (store_runs). A kernel of its own: compiled into voxelize_kernel<32, ...>
// the extra write-out path costs the aligned-grid kernels six more spilled registers and 0.5 % of the headline rate.
template <bool GAUSS, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 8 : BIG_WAVES_PER_SIMD))
    voxelize_runs_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                         const uint2 *__restrict__ slist_ext, const double *__restrict__ Tc, const float *__restrict__ kc,
                         float *__restrict__ out, const VoxParams P) {
    typedef OpsMx32<GAUSS, false, false, true> Ops;
    constexpr int CT = 32;
    constexpr bool GROUPED = false;
#include "mvx_slab_body.inc"
}

// Narrow chunks (1 ... 16 channels) of such grids: the candidate-pair walk (OpsPair) with the run-wise write-out. Until late in
// round 4 these launches went to the per-lane-range kernels (OpsF32: one candidate and one voxel per lane step, six index
// comparisons per voxel), the only narrow ones that carried store_runs: D = 49 / 50 / 63 ran at half the rate of D = 48 / 64
// (C = 4, kernel ms: 0.238 / 0.228 / 0.200 against 0.120 / 0.099 - profiles/r04_slab_plans.txt).
template <int CT, bool GAUSS, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 8 : BIG_WAVES_PER_SIMD))
    voxelize_pair_runs_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                              const uint2 *__restrict__ slist_ext, const double *__restrict__ Tc, const float *__restrict__ kc,
                              float *__restrict__ out, const VoxParams P) {
    typedef OpsPair<CT, GAUSS, true> Ops;
    constexpr bool GROUPED = false;
#include "mvx_slab_body.inc"
}

// (Measured and removed, round 3: channel counts of 32 k + r with the remainder chunk's workgroups - CTR vector-ALU
// accumulators - in the SAME launch as the full chunks' (a kernel that branches on the chunk index into two inclusions of
// the slab body). Bit-identical, and slower than the second launch it replaced: C = 33 0.618 against 0.580 ms per 64
// molecules, C = 40 0.647 / 0.605, C = 48 0.708 / 0.652, C = 65 0.967 / 0.912 - the remainder's workgroups take slots from
// the store-streaming ones for longer than their own launch lasts.)


// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
struct KernelKey {
    int ct;
    bool gauss, lane_range;
    int maxt;
};

// Calls fn.template operator()<CT, GAUSS, LANE_RANGE, MAXT>() for the instantiation `k` names.
template <typename Fn>
static hipError_t for_kernel(const KernelKey &k, Fn &&fn) {
#define MVX_CASE(CT_, G_, LR_, MT_) \
    if (k.ct == CT_ && k.gauss == G_ && k.lane_range == LR_ && k.maxt == MT_) return fn.template operator()<CT_, G_, LR_, MT_>();
#define MVX_CASES_CT(CT_, MT_)      \
    MVX_CASE(CT_, true, false, MT_)  \
    MVX_CASE(CT_, false, false, MT_) \
    MVX_CASE(CT_, true, true, MT_)   \
    MVX_CASE(CT_, false, true, MT_)
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

template <int CT, bool GAUSS, bool LANE_RANGE>
constexpr bool mx_kernel() {
    return std::is_same<typename SlabOps<CT, GAUSS, LANE_RANGE, false>::type, OpsMx32<GAUSS, LANE_RANGE, false>>::value;
}

struct GroupedFn {
    const VoxArgs &a;
    int32_t nb;
    hipStream_t s;
    template <int CT, bool GAUSS, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        if constexpr (CT != 32) {
            return hipErrorInvalidValue;
        } else {
            VoxParams p = a.p;
            if (nb <= 0) return hipSuccess;
            if ((long long)nb * p.ncc > 65535) return hipErrorInvalidConfiguration;
            static LdsLimit raised;
            const size_t main_lds = voxelize_mx_lds_bytes(p.NW);
            p.dcap = (int32_t)main_lds; // where the slots' {T, k} table sits in LDS
            auto kern = &voxelize_kernel<CT, GAUSS, LANE_RANGE, MAXT, true>;
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
    template <int CT, bool GAUSS, bool LANE_RANGE, int MAXT>
    hipError_t operator()() const {
        const VoxParams &p = a.p;
        if (nb <= 0) return hipSuccess;
        if ((long long)nb * p.ncc > 65535) return hipErrorInvalidConfiguration;
        static LdsLimit raised;
        constexpr bool mx = mx_kernel<CT, GAUSS, LANE_RANGE>();
        size_t lds = mx ? voxelize_mx_lds_bytes(p.NW) : voxelize_lds_bytes(CT, p.NW, CR_F32);
        if (CT < 32 && !LANE_RANGE) lds = std::max(lds, (size_t)65 * cand_stride_words(CT) * 4); // (the vector staging's dump row: stage_round_v)
        if constexpr (CT < 16 && !LANE_RANGE) { // narrow chunks: several sub-tiles per wave (voxelize_narrow_kernel)
            // four where the accumulator sets fit (1 or 4 channels) and the row is a multiple of four sub-tiles, else two
            const int nsub = a.narrow_sub > 0 ? a.narrow_sub : ((CT <= 4 && p.NW % 4 == 0) ? 4 : 2);
            if (nsub > 1 && p.NW % nsub == 0 && p.vec_store) {
                static LdsLimit raised_n;
                const size_t lds_n = lds; // (64 rows + the staging's dump row)
                auto launch_n = [&](auto kn) {
                    hipError_t en = raise_lds_limit(kn, lds_n, raised_n);
                    if (en != hipSuccess) return en;
                    launch_profiled(kn, dim3(slab_grid_x(p), (unsigned)(nb * p.ncc)), dim3(p.NW / nsub * 64), lds_n, s, a.rec, a.w, a.slist,
                                    a.slist_ext, static_cast<float *>(a.out), a.p);
                    return hipGetLastError();
                };
                if constexpr (CT <= 4) {
                    if (nsub == 4) return launch_n(&voxelize_narrow_kernel<CT, GAUSS, 4>);
                }
                if (nsub == 2) return launch_n(&voxelize_narrow_kernel<CT, GAUSS, 2>);
            }
        }
        auto kern = &voxelize_kernel<CT, GAUSS, LANE_RANGE, MAXT>;
        LdsLimit *state = &raised;
        if (!p.vec_store) {
            if constexpr (mx) {
                static LdsLimit raised_runs;
                kern = &voxelize_runs_kernel<GAUSS, MAXT>;
                state = &raised_runs;
            } else if constexpr (!LANE_RANGE) { // narrow chunks: the pair walk with store_runs
                static LdsLimit raised_pair_runs;
                kern = &voxelize_pair_runs_kernel<CT, GAUSS, MAXT>;
                state = &raised_pair_runs;
            }
        }
        hipError_t e = raise_lds_limit(kern, lds, *state);
        if (e != hipSuccess) return e;
        launch_profiled(kern, dim3(slab_grid_x(p), (unsigned)(nb * p.ncc)), dim3(p.NW * 64), lds, s, a.rec, a.w,
                        a.slist, a.slist_ext, a.Tc, a.kc, static_cast<float *>(a.out), a.p);
        return hipGetLastError();
    }
};

hipError_t launch_voxelize(const VoxArgs &a, int32_t nb, int32_t ct, bool gauss, bool lane_range, hipStream_t s) {
    KernelKey k{ct, gauss, lane_range, a.p.NW <= 8 ? 512 : 1024};
    return for_kernel(k, LaunchFn{a, nb, s});
}

hipError_t launch_voxelize_grouped(const VoxArgs &a, int32_t nb, bool gauss, bool lane_range, hipStream_t s) {
    KernelKey k{32, gauss, lane_range, a.p.NW <= 8 ? 512 : 1024};
    return for_kernel(k, GroupedFn{a, nb, s});
}

} // namespace mvx
