// mvx_pair.hip - voxelize_pair_kernel: a whole per-molecule forward() call in ONE launch, two slabs per workgroup (gfx950).
//
// The reference's unit of work is one molecule per forward() call (test/test_time_numpy.py:11-15; Voxelizer.forward_features /
// forward_types / forward_single, numpy/voxelizer.py:97-169, 240-315, 370-436). Such a call is a chain of latencies, not
// work: launch -> coordinates -> which atoms reach this slab -> their records and weights -> walk -> stores, in front of
// ~5.5 us of HBM drain for a cfg-2 grid. This kernel replaced the round-2 voxelize_direct_kernel (one slab per
// workgroup, every wave scanning for itself) on that path: cfg-2 call 20.5 -> 14.9 us, the reference's timing loop 17.9 -> 12.9
// (profiles/r04_single_calls.txt). What the old kernel's phase timeline showed (tools/direct_timeline.py, cfg-2: scan 12.0 of
// a workgroup's 25.6 kcycles, 21 of 27 on the reference's own timing loop) and the rules this one is built on:
//   * Sixteen waves run the front side by side on one compute unit, so every instruction of it costs ~16 cycles of the call:
//     the front is bound by its INSTRUCTION COUNT per compute unit, not by memory latency (stashing data to save a second
//     trip to memory, prefetching rows during the scan: both measured slower until the instruction count came down).
//   * A workgroup owns TWO slabs side by side along x (4 x 4 x 8 NW voxels, 2 NW waves = up to 1024 threads, one workgroup
//     per compute unit): one scan, one staged set of rows for both - half the scan work of a workgroup per slab.
//   * Scan: every load of a wave's share is issued before anything else (16-byte loads, 32-bit offsets from one scalar
//     base, 3 per 128 atoms), the box constants are computed under them, the test is ~30 instructions per 64 atoms. What
//     the scan held about a survivor (float64 position, atom-wise radius, type) stays in LDS: no second trip to memory.
//   * Stage: the survivors of all waves are gathered into rows in candidate (= atom) order - prefix over the waves by DPP -
//     and the FIRST one to four waves prepare them, one lane per row, all 64 lanes busy: the exact float64 preparation is
//     issued once or twice per workgroup instead of sixteen times with ~4 active lanes. The other waves set up their walk
//     (voxel centres, accumulators, filter constants) meanwhile.
//   * Walk and write-out are the batched kernels' (OpsMx32 on the matrix cores for 32-channel chunks, OpsPair below that),
//     so the sums are the same float32 chains in atom order: bit-identical to every other route.
//   * No scratch: the rare rounds / segments that run with the accumulators alive use lighter variants of scan and stage
//     (one block in flight, no per-lane row prefetch). A per-wave private segment of a few hundred bytes throttles the waves
//     a launch may have in flight (an in-kernel loop over channel chunks that needed 452 B: cfg-2 kernel 13.8 -> 18.0 us).
// Exactness is unchanged: the float32 scan only decides which atoms are LOOKED AT (a superset, error-bounded); box cull,
// block culls, threshold and coefficient are decided in float64 with the reference's comparisons (stage), membership per
// voxel by d2 <= T (walk).
// LDS map (dynamic): u16 list[2 NW][segw] | PairStash stash[2 NW][48] | int wcnt[32] | float rtab[256] |
//                    union { float64 strips 2 NW x 3 KB ; rows 256 x SW words ; 2 tiles }.
#include "mvx_device.h"
#include "mvx_ops32.h"

#include <algorithm>

namespace mvx {

constexpr int PAIR_BLOCK = 128;                           // atoms per transposition block (3 x 1 KB of coordinates)
constexpr int PAIR_MAX_BLOCKS = 4;                        // blocks a wave fetches at once (all loads in flight: 48 registers)
constexpr int PAIR_SEGW_MIN = PAIR_BLOCK * PAIR_MAX_BLOCKS; // atoms per wave and segment: 512 (molecules of up to 8 192 atoms on
constexpr int PAIR_SEGW_MAX = 2048;                         // 16 waves) ... 2 048 (32 768 atoms), chosen at launch (VoxParams::dcap)
constexpr int PAIR_ROWS = 256;                            // candidate rows staged per round (both slabs share them), 64 per staging wave:
                                                          // a pair of slabs at cfg-2's density holds ~62 candidates at radius 1 A, ~140 at 2 A;
                                                          // more than a round's rows means further, slower rounds (with 128 rows per round a
                                                          // molecule of 12 000 atoms took 27.9 us, with 256 23.0)
constexpr int PAIR_STASH = 48;                            // survivors per wave and segment whose float64 position, radius and type stay in LDS
constexpr int PAIR_RTAB = 256;                            // per-type radii kept in LDS (forward_types with channel-wise radii)
struct __attribute__((aligned(16))) PairStash {           // what the scan already held about a survivor: no second trip to memory
    double x, y, z;
    float r;      // atom-wise radius
    int32_t type; // forward_types channel
};
static_assert(sizeof(PairStash) == 32, "two 16-byte LDS writes per survivor");

// (the write-outs carry the run-wise path - store_runs - beside the 16-byte one: grids whose rows are not whole quads, odd
// dimensions and unaligned slices of a batch grid, take this kernel too)
// LR: blockdims whose reference blocks cut through sub-tiles (4, 5, 12, ...) - every lane checks its voxel's index against the
// atom's admitted ranges (prep_atom's block_interval), one voxel per lane and candidate (OpsF32 / OpsMx32 with per-lane ranges)
template <int CT, bool GAUSS, bool LR>
struct PairOps {
    typedef OpsPair<CT, GAUSS, true> type;
};
template <bool GAUSS>
struct PairOps<32, GAUSS, false> {
    typedef OpsMx32<GAUSS, false, false, true> type;
};
template <int CT, bool GAUSS>
struct PairOps<CT, GAUSS, true> {
    typedef OpsF32<CT, GAUSS, true> type;
};
template <bool GAUSS>
struct PairOps<32, GAUSS, true> {
    typedef OpsMx32<GAUSS, true, false, true> type;
};

__host__ __device__ inline int pair_tile_words(int ct, int NW) { // one slab's write-out tile (Ops::write)
    const int cr = ct == 32 ? MX_CR : (ct < CR_F32 ? ct : CR_F32);
    return cr * RPC * row_stride_floats(NW);
}
// atoms per wave and segment for molecules of up to max_atoms atoms: one segment whenever the list can hold it (a second
// segment repeats scan, barriers, stage and walk with the accumulators alive: 12 000 atoms took 37.8 us in two segments of
// 8 192 against 30.6 us binned)
static int32_t pair_segw(int64_t max_atoms, int32_t NW) {
    int64_t per_wave = (max_atoms + 2 * NW - 1) / (2 * NW);
    per_wave = (per_wave + PAIR_BLOCK - 1) / PAIR_BLOCK * PAIR_BLOCK;
    return (int32_t)std::min<int64_t>(std::max<int64_t>(per_wave, PAIR_SEGW_MIN), PAIR_SEGW_MAX);
}
__host__ __device__ inline int pair_rows(int NW) { return PAIR_ROWS < 128 * NW ? PAIR_ROWS : 128 * NW; } // (64 per wave of the workgroup at most)
static size_t pair_lds_bytes(int32_t ct, int32_t NW, int32_t segw) {
    const size_t strips = (size_t)2 * NW * PAIR_BLOCK * 24;
    const size_t rows = (size_t)pair_rows(NW) * cand_stride_words(ct) * 4;
    const size_t tiles = (size_t)2 * pair_tile_words(ct, NW) * 4;
    const size_t un = std::max(strips, std::max(rows, tiles));
    return (size_t)2 * NW * ((size_t)segw * 2 + PAIR_STASH * sizeof(PairStash)) + 128 + PAIR_RTAB * 4 + un;
}

typedef unsigned u4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f4a16 __attribute__((ext_vector_type(4)));

// inclusive prefix over the 16 lanes of a row (row_shr: lanes shifted in from outside the row read 0)
__device__ __forceinline__ int row_prefix16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}

template <int CT, bool GAUSS, bool XF, bool LR = false>
__global__ void __launch_bounds__(1024) voxelize_pair_kernel(const DirectArgs A, float *__restrict__ out, const VoxParams P) {
    typedef typename PairOps<CT, GAUSS, LR>::type Ops;
    constexpr int SW = Ops::SW;
    constexpr int WW = Ops::WW; // weight words per row
    // blocks of 128 atoms a wave fetches at once: four (48 registers of coordinates in flight), three where the transform's
    // constants share the register file (with four the scan of the transform variants kept a dozen registers in scratch:
    // the reference's timing loop 12.4 -> 15.1 us per call)
    constexpr int HOTB = XF ? 3 : PAIR_MAX_BLOCKS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = P.NW, NWT = 2 * NW;
    const int h = wave >= NW ? 1 : 0; // which slab of the pair
    const int ws = wave - h * NW;     // this wave's sub-tile along z
    unsigned short *list = reinterpret_cast<unsigned short *>(smem);
    const int SEGW = P.dcap; // atoms per wave and segment (pair_segw)
    PairStash *stash_all = reinterpret_cast<PairStash *>(smem + (size_t)NWT * SEGW * 2);
    int *wcnt = reinterpret_cast<int *>(stash_all + (size_t)NWT * PAIR_STASH);
    float *rtab = reinterpret_cast<float *>(wcnt + 32);
    unsigned *un = reinterpret_cast<unsigned *>(rtab + PAIR_RTAB);
    unsigned short *region = list + wave * SEGW; // this wave's survivors, in atom order
    PairStash *stash = stash_all + wave * PAIR_STASH; // ... and what the scan knew about the first PAIR_STASH of them

    int b = (int)blockIdx.y, cc = 0;
    if (P.ncc > 1) { // (channel chunks - C > 32 - are separate workgroups: an in-kernel loop cost more than it shared, see above)
        b = (int)__umulhi(blockIdx.y, P.ncc_inv); // blockIdx.y / ncc
        cc = (int)blockIdx.y - b * P.ncc;
    }
    int px, sy, zc;
    decode_slab(blockIdx.x, P, px, sy, zc); // (nzc == 1: pair id = sy + nsy * px)
    const int x0p = 2 * SUBX * px, x0 = x0p + SUBX * h, y0 = SUBY * sy, z0 = 0;
    const int cbase = cc * CT;
    const PrepArgs &pa = A.pa;
    const int C = pa.C;
    const Geom &g = pa.g;
    const int D = P.D;

    int64_t a0 = 0, a1 = A.N;
    if (pa.offsets) {
        a0 = pa.offsets[b];
        a1 = pa.offsets[b + 1];
    }
    const int N = (int)(a1 - a0);
    mvx_xform xf;
    if constexpr (XF) {
        xf = pa.xf_one;
        if (pa.xforms) xf = pa.xforms[b];
        // a device-resident centre (the reference's timing loop hands `center` over as a tensor): fetched ONCE, through the
        // scalar cache, instead of by a vector load in front of the scan's constants and another in front of the stage's
        // float64 transform (each a dependent trip to L2 on the critical path). The values were written before this launch.
        if (xf.flags & MVX_XF_CENTER_PTR) {
            typedef const double __attribute__((address_space(4))) *const_f64;
            const const_f64 cp = (const_f64)(reinterpret_cast<uintptr_t>(xf.center_ptr));
            xf.center[0] = cp[0];
            xf.center[1] = cp[1];
            xf.center[2] = cp[2];
            xf.flags &= ~(uint32_t)MVX_XF_CENTER_PTR;
        }
    }
#ifdef MVX_DIAG // per-workgroup s_memtime stamps into the (otherwise unused) record buffer: diagnostic builds only
    // (16 slots per workgroup = the 8 per slab the host allocates, zeroed before the launch; slots 11-13 take the LATEST wave's time)
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(pa.rec) + 16 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#define MVX_STAMP(i) do { if (tid == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MVX_STAMP_MAX(i) do { if (lane == 0) atomicMax(&stamps[i], (unsigned long long)__builtin_amdgcn_s_memtime()); } while (0)
    MVX_STAMP_MAX(13);
#else
#define MVX_STAMP(i) do { } while (0)
#define MVX_STAMP_MAX(i) do { } while (0)
#endif
    MVX_STAMP(0);

    const int ROWS = pair_rows(NW); // candidate rows per round
    const bool small = N <= ROWS; // ligands: no scan - every atom gets a row, the stage's own tests drop the far ones
    const int SEGN = NWT * SEGW;

    // ---- A. scan of the segment that starts at atom s0 of the molecule: this wave's survivors -> region[0 .. cnt) ------------
    // (deliberately short: sixteen waves run it side by side, so every instruction here costs ~16 cycles of a call)
    auto scan = [&](int s0, auto blocks_tag) __attribute__((always_inline)) -> int {
        constexpr int MAXB = decltype(blocks_tag)::value; // blocks fetched at once (fewer where the accumulators are alive)
        MVX_STAMP(8);
        const int nseg = (N - s0) < SEGN ? (N - s0) : SEGN; // atoms of this segment
        const int bpw = (nseg + PAIR_BLOCK * NWT - 1) / (PAIR_BLOCK * NWT); // blocks per wave, 1 ... SEGW / 128
        const int wbeg = wave * bpw * PAIR_BLOCK; // first atom of this wave's share, relative to the segment
        int cnt = 0;
        if (wbeg >= nseg) return 0;
        // every load of this wave's share is issued before anything else: lane l holds bytes [16 l, 16 l + 16) of each
        // 1-KB third of a 128-atom block (clamped at the molecule's last 16 bytes: atoms past the end are masked), plus the
        // atoms' radii / types where the call has them per atom
        const char *cb = reinterpret_cast<const char *>(pa.coords + 3 * (a0 + s0));
        const unsigned lim = 24u * (unsigned)(N - s0) - 16u;
        const bool per_atom = pa.radii_src == RAD_ATOM;
        const bool typed = pa.types != nullptr;
        u4a8 fd[MAXB][3];
        float fr[MAXB][2];
        int ft[MAXB][2];
        auto issue = [&](int b0) __attribute__((always_inline)) { // the loads of blocks b0 .. b0 + 3 of this wave's share
#pragma unroll
            for (int blk = 0; blk < MAXB; ++blk) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fr[blk][q] = 0.0f;
                    ft[blk][q] = 0;
                }
                if (b0 + blk < bpw) {
                    const unsigned ob = 24u * (unsigned)(wbeg + (b0 + blk) * PAIR_BLOCK) + 16u * (unsigned)lane;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const unsigned o = ob + 1024u * k;
                        u4a8 v = *reinterpret_cast<const u4a8 *>(cb + (o < lim ? o : lim));
                        // an odd atom count ends in the middle of a 16-byte chunk: that chunk is fetched 8 bytes early (never
                        // a byte past the molecule), so the last coordinate arrives in the upper half
                        if (o == lim + 8u) {
                            v.x = v.z;
                            v.y = v.w;
                        }
                        fd[blk][k] = v;
                    }
                    if (per_atom | typed) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int a = wbeg + (b0 + blk) * PAIR_BLOCK + 64 * q + lane;
                            const int64_t ag = a0 + s0 + (a < nseg ? a : nseg - 1);
                            if (per_atom) fr[blk][q] = static_cast<const float *>(pa.radii)[ag];
                            if (typed) ft[blk][q] = pa.types[ag];
                        }
                    }
                }
            }
        };
        issue(0);
        MVX_STAMP(9);
        // (under the loads) the pair's box as float32 centre and half extents: voxels x0p .. x0p + 3, y0 .. y0 + 3, whole
        // rows (a pair that sticks out of the grid - D % 4 != 0 - is tested with its full box: still a superset), minus the
        // transform's final offset; the half extents carry the
        // rounding of this very estimate (a few 1e-7 of the magnitudes involved). The atom's side of the error bound is
        // SCAN_MARGIN times its magnitude (make_xform_f32 in mvx_device.h); membership is never decided here.
        const float resf = (float)P.res, halff = (float)P.half;
        XformF32 X;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
        if constexpr (XF) {
            X = make_xform_f32(xf);
            o0 = X.o0;
            o1 = X.o1;
            o2 = X.o2;
        }
        const float eps = 4.0e-6f * (fabsf(halff) + resf * (float)D + fabsf(o0) + fabsf(o1) + fabsf(o2) + 1.0f);
        const float ccx = ((float)x0p + 1.5f) * resf - halff - o0, ccy = ((float)y0 + 1.5f) * resf - halff - o1,
                    ccz = 0.5f * (float)(D - 1) * resf - halff - o2;
        const float hx = 1.5f * resf + eps, hy = hx, hz = 0.5f * (float)(D - 1) * resf + eps;
        // radius window of the scan: the scalar radius; per-type radii: the largest usable one of the table (the exact
        // radius is the stage's business); atom-wise radii: fetched with the coordinates
        float rwin_u = 0.0f;
        if (pa.radii_src == RAD_SCALAR) rwin_u = (float)pa.radius_scalar;
        else if (pa.radii_src == RAD_CHANNEL_BY_TYPE) {
            float m = 0.0f;
            for (int c = lane; c < C; c += 64) {
                const float r = static_cast<const float *>(pa.radii)[c];
                if (r > m && r < 3.0e38f) m = r;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            rwin_u = m;
        }
        // rr = rwin * 1.00001 + SCAN_MARGIN * mag + 1e-6 with mag = scale * (|x| + |y| + |z|) + X.mag (1 without a transform)
        float mscale = SCAN_MARGIN, mbase = SCAN_MARGIN + 1e-6f;
        if constexpr (XF) {
            mscale = SCAN_MARGIN * X.scale;
            mbase = SCAN_MARGIN * X.mag + 1e-6f;
        }
        const float rbase_u = rwin_u * 1.00001f + mbase;
        double *strip = reinterpret_cast<double *>(un) + (size_t)wave * PAIR_BLOCK * 3; // this wave's transposition strip
#pragma nounroll
        for (int b0 = 0;;) { // (one trip for shares of up to 512 atoms: molecules of up to 8 192 atoms on sixteen waves)
    #pragma unroll
            for (int blk = 0; blk < MAXB; ++blk) {
                if (b0 + blk < bpw) {
    #pragma unroll
                    for (int k = 0; k < 3; ++k)
                        *reinterpret_cast<uint4 *>(strip + 128 * k + 2 * lane) = make_uint4(fd[blk][k].x, fd[blk][k].y, fd[blk][k].z, fd[blk][k].w);
                    if (b0 + blk == 0) MVX_STAMP(10);
    #pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int j = 64 * q + lane;
                        const int a = wbeg + (b0 + blk) * PAIR_BLOCK + j;
                        const double xd = strip[3 * j], yd = strip[3 * j + 1], zd = strip[3 * j + 2];
                        float x = (float)xd, y = (float)yd, z = (float)zd;
                        const float asum = fabsf(x) + fabsf(y) + fabsf(z);
                        const float rr = mscale * asum + (per_atom ? fr[blk][q] * 1.00001f + mbase : rbase_u);
                        if constexpr (XF) {
                            x -= X.c0;
                            y -= X.c1;
                            z -= X.c2;
                            if (X.rot) {
                                const float u = X.m00 * x + X.m01 * y + X.m02 * z;
                                const float v = X.m10 * x + X.m11 * y + X.m12 * z;
                                const float w = X.m20 * x + X.m21 * y + X.m22 * z;
                                x = u;
                                y = v;
                                z = w;
                            }
                        }
                        // every test widened by the estimate's error bound; magnitudes float32 cannot hold are left to float64
                        const bool near = (fabsf(x - ccx) <= hx + rr) & (fabsf(y - ccy) <= hy + rr) & (fabsf(z - ccz) <= hz + rr);
                        const bool ok = (a < nseg) & (near | !(asum < 1.0e30f));
                        const unsigned long long mk = __ballot(ok);
                        if (ok) {
                            const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                            region[pos] = (unsigned short)a;
                            if (pos < PAIR_STASH) {
                                typedef double d2v __attribute__((ext_vector_type(2)));
                                d2v *e = reinterpret_cast<d2v *>(stash + pos);
                                e[0] = (d2v){xd, yd};
                                e[1] = (d2v){zd, __hiloint2double(ft[blk][q], (int)__float_as_uint(fr[blk][q]))};
                            }
                        }
                        cnt += __popcll(mk);
                    }
                }
            }
            b0 += MAXB;
            if (b0 >= bpw) break;
            issue(b0);
        }
        MVX_STAMP_MAX(11);
        return cnt;
    };

    // this wave's survivors and where they fall in the segment's candidate order (one barrier)
    auto prefix = [&](int cnt, int &pre_mine) __attribute__((always_inline)) -> int {
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        const int v = row_prefix16(lane < NWT ? wcnt[lane] : 0); // (lanes 0-15: inclusive prefix over the waves)
        pre_mine = wave ? __builtin_amdgcn_readlane(v, wave - 1) : 0;
        MVX_STAMP(1);
        return __builtin_amdgcn_readlane(v, 15);
    };
    // ---- B1. gather: survivors [.., ..) of this wave that fall into the round that starts at candidate r0 -> the round's rows,
    //          in candidate (= atom) order: words 0-5 position, 6 atom-wise radius, 7 type, 13 atom index (bit 31: not stashed)
    auto gather = [&](int cnt, int pre, int r0) __attribute__((always_inline)) {
        for (int i = lane; i < cnt; i += 64) {
            const int rw = pre + i - r0;
            if (rw >= 0 && rw < ROWS) {
                unsigned *row = un + (size_t)rw * SW;
                if (i < PAIR_STASH) {
                    const uint4 *e = reinterpret_cast<const uint4 *>(stash + i);
                    reinterpret_cast<uint4 *>(row)[0] = e[0];
                    reinterpret_cast<uint4 *>(row)[1] = e[1];
                }
                row[13] = (unsigned)region[i] | (i < PAIR_STASH ? 0u : 0x80000000u);
            }
        }
    };
    // ---- B2. stage: the first n / 64 waves turn the round's n rows into records + channel weights, one lane per row (all 64 lanes
    //          busy: the exact float64 preparation is issued once or twice per workgroup, not once per scanning wave)
    auto stage = [&](int s0, int n, auto own_tag) __attribute__((always_inline)) {
        constexpr bool OWN = decltype(own_tag)::value; // feature rows fetched by the row's own lane (CT / 4 x 4 registers in flight)
        const int rw = 64 * wave + lane;
        if (64 * wave < n) {
            const bool valid = rw < n;
            unsigned *row = un + (size_t)rw * SW;
            unsigned arel = 0; // atom index inside the molecule
            if (valid) {
                double p[3];
                int my_type = 0;
                float r_atom = 0.0f;
                bool stashed = false;
                if (small) arel = (unsigned)rw;
                else {
                    const unsigned tag = row[13];
                    arel = (unsigned)s0 + (tag & 0xffffu);
                    stashed = (tag >> 31) == 0u;
                }
                if (stashed) {
                    p[0] = *reinterpret_cast<const double *>(row);
                    p[1] = *reinterpret_cast<const double *>(row + 2);
                    p[2] = *reinterpret_cast<const double *>(row + 4);
                    r_atom = __uint_as_float(row[6]);
                    my_type = (int)row[7];
                } else { // molecules that skip the scan, survivors beyond the stash: a trip to memory
                    const double *cp = pa.coords + 3 * a0;
                    p[0] = cp[3u * arel];
                    p[1] = cp[3u * arel + 1u];
                    p[2] = cp[3u * arel + 2u];
                    if (pa.types) my_type = (pa.types + a0)[arel];
                    if (pa.radii_src == RAD_ATOM) r_atom = (static_cast<const float *>(pa.radii) + a0)[arel];
                }
                bool keep = true;
                if (pa.types && (my_type < 0 || my_type >= C)) keep = false;
                float r32;
                double rc;
                if (pa.radii_src == RAD_SCALAR) {
                    rc = pa.radius_scalar;
                    r32 = (float)pa.radius_scalar;
                } else {
                    if (pa.radii_src == RAD_ATOM) r32 = r_atom;
                    else r32 = keep ? ((!small && my_type < PAIR_RTAB) ? rtab[my_type] : static_cast<const float *>(pa.radii)[my_type]) : 0.0f; // numpy/voxelizer.py:284-285
                    rc = (double)r32;
                }
                // features whose rows are whole 16-byte quads: this lane fetches its own row (CT / 4 loads in flight)
                f4a16 wq[CT >= 4 ? CT / 4 : 1];
                const bool own_row = OWN && pa.mode == MODE_FEATURES && CT >= 4 && (C & 3) == 0 && cbase + CT <= C &&
                                     (reinterpret_cast<uintptr_t>(pa.features) & 15u) == 0;
                if (own_row) {
                    const f4a16 *fp = reinterpret_cast<const f4a16 *>(static_cast<const float *>(pa.features) + (a0 + arel) * C + cbase);
#pragma unroll
                    for (int qd = 0; qd < CT / 4; ++qd) wq[qd] = fp[qd];
                }
#ifdef MVX_DIAG
                if (p[0] != 1.2345e300) MVX_STAMP(7); // (after the coordinates have arrived)
#endif
                if constexpr (XF) apply_xform(xf, p[0], p[1], p[2]);
                if constexpr (LR) {
                    // per-lane voxel ranges are needed: the prep kernel's own code for the whole record (it fetches type and
                    // radius itself), the pair's box against the admitted ranges
                    AtomRec R;
                    uint32_t rng[3];
                    bool keepr = prep_atom(pa, a0 + (int64_t)arel, p, 0.0f, 0.0, R, rng);
                    my_type = R.type;
                    keepr = keepr && ((int)(rng[0] & 0xffff) <= x0p + 2 * SUBX - 1) && ((int)(rng[0] >> 16) >= x0p) &&
                            ((int)(rng[1] & 0xffff) <= y0 + SUBY - 1) && ((int)(rng[1] >> 16) >= y0);
                    if (!keepr) R.xr = R.yr = R.zr = EMPTY_RANGE;
                    const uint4 *src = reinterpret_cast<const uint4 *>(&R);
                    uint4 *dst = reinterpret_cast<uint4 *>(row);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) dst[q4] = src[q4];
                } else {
                    const double ub = g.half, lb = -1 * g.half;
                    if (pa.radii_src == RAD_SCALAR) {
                        for (int q = 0; q < 3; ++q) keep = keep && (p[q] > lb - rc) && (p[q] < ub + rc); // numpy/voxelizer.py:487-488
                    } else {
                        for (int q = 0; q < 3; ++q) keep = keep && (p[q] + rc > lb) && (p[q] - rc < ub); // :491-492
                    }
                    // (one python float for every atom: threshold and coefficient come with the launch)
                    const double T = pa.radii_src == RAD_SCALAR ? pa.T_scalar : d2_threshold(r32);
                    keep = keep && (T >= 0.0);
                    // sub-tiles lie inside one reference block: the x / y block culls are uniform over the pair (its 4 x 4 voxels
                    // share a block: blockdim is a multiple of 8 here), the z cull over each wave's sub-tile (walk)
                    const BlockBounds Bx = block_bounds_lane(g, x0p), By = block_bounds_lane(g, y0);
                    keep = keep && block_admits(Bx, p[0], rc) && block_admits(By, p[1], rc);
                    // the pair's box (voxel centres), widened by the atom's radius window (conservative, as prep_atom's)
                    const double slack = 1e-6 * P.res;
                    const double bx0 = (double)x0p * P.res - P.half - slack, bx1 = (double)(x0p + 2 * SUBX - 1) * P.res - P.half + slack;
                    const double by0 = (double)y0 * P.res - P.half - slack, by1 = (double)(y0 + SUBY - 1) * P.res - P.half + slack;
                    const double bz0 = -P.half - slack, bz1 = (double)(D - 1) * P.res - P.half + slack;
                    const double rrd = (double)r32 * 1.000001 + 1e-9;
                    keep = keep && (p[0] + rrd >= bx0) && (p[0] - rrd <= bx1) && (p[1] + rrd >= by0) && (p[1] - rrd <= by1) &&
                           (p[2] + rrd >= bz0) && (p[2] - rrd <= bz1);
                    typedef double d2v __attribute__((ext_vector_type(2)));
                    d2v *dst = reinterpret_cast<d2v *>(row);
                    dst[0] = (d2v){p[0], p[1]};
                    dst[1] = (d2v){p[2], T};
                    row[8] = __float_as_uint(!GAUSS ? 0.0f : (pa.radii_src == RAD_SCALAR ? pa.k_scalar : gauss_coeff(r32, pa.sigma32)));
                    row[9] = (unsigned)my_type;
                    *reinterpret_cast<double *>(row + 10) = rc;
                    // window radius, rounded up to float; a dropped candidate gets a negative one
                    row[12] = __float_as_uint(keep ? (float)rrd * 1.0000002f : -1.0f);
                }
                if (own_row) {
#pragma unroll
                    for (int qd = 0; qd < CT / 4; ++qd) *reinterpret_cast<f4a16 *>(row + 16 + 4 * qd) = wq[qd];
                } else if (pa.mode != MODE_FEATURES) { // one-hot type row / the unit weight of forward_single
#pragma unroll
                    for (int j = 0; j < WW; ++j) {
                        const bool one = pa.mode == MODE_TYPES ? (my_type == cbase + j) : (j == 0);
                        row[16 + j] = one ? 0x3f800000u : 0u;
                    }
                }
            }
            if (pa.mode == MODE_FEATURES) { // (uniform) any other feature layout: 64 / WW rows per load, one word per lane
                const bool own_row = OWN && CT >= 4 && (C & 3) == 0 && cbase + CT <= C && (reinterpret_cast<uintptr_t>(pa.features) & 15u) == 0;
                if (!own_row) {
                    constexpr int RPI = 64 / WW;
                    const int nw = (n - 64 * wave) < 64 ? (n - 64 * wave) : 64; // rows of this wave
                    const float *feat = static_cast<const float *>(pa.features) + a0 * C;
                    for (int q0 = 0; q0 < nw; q0 += RPI) {
                        const int rr_ = q0 + lane / WW, j = lane % WW;
                        const unsigned ar = (unsigned)__shfl((int)arel, rr_ < 64 ? rr_ : 63); // the atom of row 64 wave + rr_
                        if (rr_ < nw) {
                            float v = 0.0f;
                            if (cbase + j < C) v = feat[(size_t)ar * C + cbase + j];
                            un[(size_t)(64 * wave + rr_) * SW + 16 + j] = __float_as_uint(v);
                        }
                    }
                }
            }
        }
        MVX_STAMP(2);
        MVX_STAMP_MAX(12);
    };

    // (voxel centres, accumulators and the row filter's per-wave constants are set up by set_walk(), which every wave calls
    // between its share of the staging and the barrier in front of the first walk: the waves that have nothing to stage do it
    // while the first one to four waves prepare the records)
    LaneCtx L;
    typename Ops::Acc acc;
    bool any = false;
    BlockBounds Bz;
    double wz0, wz1, wx0, wx1;
    const int zv = z0 + SUBZ * ws; // first voxel of this wave's sub-tile
    auto set_walk = [&]() __attribute__((always_inline)) {
        L = Ops::ctx(lane, ws, x0, y0, z0, 0, cbase, P);
        Ops::zero(acc);
        const double slack = 1e-6 * P.res;
        const int zl = (zv + SUBZ - 1 < D - 1) ? zv + SUBZ - 1 : D - 1;
        // (wave-uniform: kept in scalar registers across the walk)
        Bz = block_bounds_lane(g, zv);
        Bz.lo = uniform(Bz.lo);
        Bz.hi = uniform(Bz.hi);
        wz0 = uniform((double)zv * P.res - P.half - slack);
        wz1 = uniform((double)zl * P.res - P.half + slack);
        wx0 = uniform((double)x0 * P.res - P.half - slack);
        wx1 = uniform((double)(x0 + SUBX - 1) * P.res - P.half + slack);
    };
    // ---- C. the rows this wave's sub-tile takes (one lane per row), then the walk ----------------------------------------
    auto walk = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < PAIR_ROWS / 64; ++half) {
            if (64 * half < n) {
                bool ok = false, kept = false;
                const int rw = 64 * half + lane;
                if (rw < n) {
                    const unsigned *r = un + (size_t)rw * SW;
                    if constexpr (LR) { // the record's admitted voxel ranges against this wave's sub-tile
                        const unsigned xr = r[10], yr = r[11], zr = r[12];
                        kept = xr != EMPTY_RANGE;
                        ok = kept && (zv < D) && (x0 < D) && ((int)(zr & 0xffff) <= zv + SUBZ - 1) && ((int)(zr >> 16) >= zv) &&
                             ((int)(xr & 0xffff) <= x0 + SUBX - 1) && ((int)(xr >> 16) >= x0) && ((int)(yr & 0xffff) <= y0 + SUBY - 1) &&
                             ((int)(yr >> 16) >= y0);
                    } else {
                        const double pxr = *reinterpret_cast<const double *>(r);
                        const double pz = *reinterpret_cast<const double *>(r + 4);
                        const double rc = *reinterpret_cast<const double *>(r + 10);
                        const double rr = (double)__uint_as_float(r[12]);
                        kept = rr >= 0.0;
                        // (zv < D: sub-tiles past the end of a row; x0 < D: the second slab of the last pair when the grid has an odd
                        // number of x-slabs - such waves walk nothing and their write-out stores nothing, but they pass every barrier)
                        ok = kept && (zv < D) && (x0 < D) && block_admits(Bz, pz, rc) && (pz + rr >= wz0) && (pz - rr <= wz1) &&
                             (pxr + rr >= wx0) && (pxr - rr <= wx1);
                    }
                }
                any = any || __ballot(kept) != 0ull; // (the same rows in every wave: workgroup-uniform)
                Ops::walk(acc, __ballot(ok), un + (size_t)64 * half * SW, lane, L, P, nullptr, nullptr);
            }
        }
        MVX_STAMP(4);
    };

    // The first round is staged BEFORE the accumulators exist: scan and stage have the whole register file, and
    // per-molecule calls rarely need more than this one round per pair.
    auto round = [&](int s0, int cnt, int pre, int total, int r0) __attribute__((always_inline)) { // (the rounds after a segment's first: cold)
        const int n = (total - r0) < ROWS ? (total - r0) : ROWS;
        __syncthreads(); // every wave is done with the previous round's rows / the segment's strips
        gather(cnt, pre, r0);
        __syncthreads();
        stage(s0, n, std::false_type{}); // (cold: no row prefetch beside the live accumulators)
        __syncthreads();
        walk(n);
    };
    if (small) { // every atom is a candidate: no scan, no list - one barrier in the whole front
        if (N > 0) stage(0, N, std::true_type{});
        set_walk();
        if (N > 0) {
            __syncthreads();
            MVX_STAMP(3);
            walk(N);
        }
    } else {
        if (pa.radii_src == RAD_CHANNEL_BY_TYPE) // per-type radii: the table into LDS (published by the prefix barrier)
            for (int c = tid; c < (C < PAIR_RTAB ? C : PAIR_RTAB); c += (int)blockDim.x) rtab[c] = static_cast<const float *>(pa.radii)[c];
        // first segment, first round (the usual whole of a call): nothing of the walk is alive yet
        const int cnt0 = scan(0, std::integral_constant<int, HOTB>{});
        int pre0;
        const int total0 = prefix(cnt0, pre0); // (barrier: every strip has been read)
        const int n0 = total0 < ROWS ? total0 : ROWS;
        if (total0 > 0) {
            gather(cnt0, pre0, 0);
            __syncthreads();
            stage(0, n0, std::true_type{});
        }
        set_walk();
        if (total0 > 0) {
            __syncthreads();
            MVX_STAMP(3);
            walk(n0);
#pragma nounroll
            for (int r0 = ROWS; r0 < total0; r0 += ROWS) round(0, cnt0, pre0, total0, r0);
        }
#pragma nounroll
        for (int s0 = SEGN; s0 < N; s0 += SEGN) { // molecules of more than 512 atoms per wave: further segments
            __syncthreads(); // rows consumed before the scan strips overwrite them
            const int cnt = scan(s0, std::integral_constant<int, 1>{}); // (cold: the accumulators are alive)
            int pre;
            const int total = prefix(cnt, pre);
#pragma nounroll
            for (int r0 = 0; r0 < total; r0 += ROWS) round(s0, cnt, pre, total, r0);
        }
    }
    MVX_STAMP(5);
    // write-out of this wave's slab (each half has its own tile; `any` is uniform over the pair, so both halves pass the
    // same barriers); begins with a barrier
    unsigned *tile = un + (size_t)h * pair_tile_words(CT, NW);
    Ops::write(acc, any ? 1 : 0, tile, tid - h * NW * 64, lane, ws, NW, b, L, x0, y0, z0, out, P);
    MVX_STAMP(6);
#undef MVX_STAMP
#undef MVX_STAMP_MAX
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <int CT, bool GAUSS, bool XF, bool LR>
static hipError_t launch_pair_t(const DirectArgs &d, const VoxParams &p, int64_t max_atoms, float *out, hipStream_t s) {
    static LdsLimit raised;
    VoxParams q = p;
    q.dcap = pair_segw(max_atoms, p.NW);
    const size_t lds = pair_lds_bytes(CT, p.NW, q.dcap);
    auto kern = &voxelize_pair_kernel<CT, GAUSS, XF, LR>;
    hipError_t e = raise_lds_limit(kern, lds, raised);
    if (e != hipSuccess) return e;
    launch_profiled(kern, dim3((unsigned)(p.nsy * ((p.nsx + 1) / 2)), (unsigned)(p.B * p.ncc)), dim3(p.NW * 128), lds, s, d, out, q);
    return hipGetLastError();
}

// The whole call in one launch: float32 grids with whole rows per slab (NW <= 8); rows of whole 16-byte quads or not (run-wise
// write-out), an even number of x-slabs or not (the last pair's second slab then lies outside the grid). lane_range: sub-tiles
// cut by reference blocks (blockdim 4, 5, 12, ...) - built with the transform-capable instantiation only (an identity
// transform costs that rare case little and keeps the number of kernels down).
hipError_t launch_voxelize_direct(const DirectArgs &d, const VoxParams &p, int64_t max_atoms, float *out, int32_t ct, bool gauss,
                                  bool lane_range, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.NW > 8 || p.nzc != 1 || (long long)p.B * p.ncc > 65535) return hipErrorInvalidConfiguration;
    if (max_atoms > (int64_t)100000000) return hipErrorInvalidConfiguration; // (24-byte rows addressed with 32-bit offsets; plan_call stops at 131 072 atoms)
    const bool xf = d.pa.xforms != nullptr || d.pa.xf_one.flags != 0;
#define MVX_CASE(CT_)                                                                                                       \
    if (ct == CT_) {                                                                                                        \
        if (lane_range)                                                                                                     \
            return gauss ? launch_pair_t<CT_, true, true, true>(d, p, max_atoms, out, s) : launch_pair_t<CT_, false, true, true>(d, p, max_atoms, out, s); \
        if (gauss)                                                                                                          \
            return xf ? launch_pair_t<CT_, true, true, false>(d, p, max_atoms, out, s) : launch_pair_t<CT_, true, false, false>(d, p, max_atoms, out, s); \
        return xf ? launch_pair_t<CT_, false, true, false>(d, p, max_atoms, out, s) : launch_pair_t<CT_, false, false, false>(d, p, max_atoms, out, s);   \
    }
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
