// mvx_ops32.h - the two float32 accumulator layouts shared by the batched slab kernels (mvx_slab.hip) and the per-molecule
// kernel (mvx_pair.hip): OpsMx32 (32-channel chunks on the matrix cores) and OpsPair (1 ... 16 channels on the vector ALU,
// candidates evaluated in pairs). Same row format (AtomRec words + weights at word 16), same arithmetic, same bits.
#pragma once
#include "mvx_device.h"

namespace mvx {

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

template <bool GAUSS, bool LANE_RANGE, bool GROUPED_ = false, bool RUNS_ = LANE_RANGE>
struct OpsMx32 {
    static constexpr int CT = 32;
    static constexpr bool RUNS = RUNS_; // carries the run-wise write-out (store_runs)
    static constexpr bool GROUPED = GROUPED_;
    static constexpr bool VSTAGE = false;
    static constexpr bool PRESTAGE = true; // the row region holds two rounds: stage_first_rounds
    static constexpr bool CULL = true;
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
    static __device__ __forceinline__ void tables(LaneCtx &, char *, const VoxParams &, int) {}
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
                // hita / hitb so far only say "inside the largest ball" (and the index ranges). Per radius slot of this
                // chunk: its own threshold and density for the same d2, the weight row masked to its channels - channels
                // of other slots receive fma(0, val, acc) = acc. Slots come by descending radius (chan_aux_kernel) and
                // thresholds grow with the radius: the first slot that no lane of the wave hits ends the candidate pair.
                const float d2fa = (float)d2a, d2fb = (float)d2b;
                for (int g = 0; g < L.nslots; ++g) {
                    const double Tg = L.gtab[2 * g];
                    const bool ha = hita && d2a <= Tg, hb = hitb && d2b <= Tg;
                    if (__ballot(ha || hb) == 0ull) break; // (uniform)
                    const float kg = reinterpret_cast<const float *>(L.gtab + 2 * g + 1)[0];
                    const float eva = GAUSS ? __builtin_amdgcn_exp2f(kg * d2fa) : 1.0f;
                    const float evb = GAUSS ? __builtin_amdgcn_exp2f(kg * d2fb) : 1.0f;
                    const float va = ha ? eva : 0.0f, vb = hb ? evb : 0.0f;
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
        const int rfirst = (int)(((float)tid + 0.5f) * __frcp_rn((float)F4)); // (= tid / F4 without the integer division: write_slab)
        const int q = tid - rfirst * F4, zq = z0 + 4 * q;
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
                        store_f4(dst0 + (size_t)(rd * CR + 4 * p) * D3, v);
                    }
                }
            }
            // Pacing (any == 2: a slab of few candidates in a store-bound launch; limits and measurements in mvx_tuning.h):
            // the wave holds back after each round's stores about as long as a compute unit needs to drain the bytes the
            // workgroup just queued (2.5 NW units of 64 cycles: 1280 cycles for 16 KB at NW = 8). Left alone a workgroup
            // pushes its 64 KB within ~2 kcycles and the row loads of the unit's other workgroups wait behind them. Same
            // box, kernel, of peak: cfg-2 x 256 0.771-0.794 -> 0.807-0.811, cfg-5 x 8 0.697 -> 0.711. Waiting for the
            // stores' acknowledgement instead (s_waitcnt vmcnt(0)) gives +4 % where the sleep gives +4.7 %; a sleep after
            // every store instruction, or waves starting their first round apart, the same or less.
            if (rd < 3 && any == 2)
                for (int i = 0; i < NW; i += 2) __builtin_amdgcn_s_sleep(ROUND_SLEEP_STEP);
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
        round(R0{});
        round(R1{});
        round(R2{});
        round(R3{});
    }
};

// ---- narrow chunks (1 ... 16 channels) on the vector ALU, two candidates per step ---------------------------------------
// forward_single, forward_types with a few element channels, cfg-1's 5 and cfg-4's 16 feature channels: little to store, so
// the walk is what such launches cost - and with one voxel per lane (OpsF32) they were bound by the compute unit's one
// scalar unit (304 scalar against 237 vector instructions per wave at C = 1, scalar issue 80 % busy:
// profiles/r04_narrow.txt) with the vector ALU close behind. OpsPair keeps OpsF32's accumulators (one voxel per lane, CT
// channels: same write-out) but EVALUATES like OpsMx32: lanes 0-31 take candidate A, lanes 32-63 candidate B of a pair, each
// lane for the two voxels (x0, ly, lz) and (x0 + 1, ly, lz) - dy^2 and dz^2 shared: 14 float64 operations per pair
// instead of 18, one trip of the scalar loop per pair instead of two. v_permlane32_swap then hands every lane the two
// candidates' values at its OWN voxel (lanes 32-63 own the x0 + 1 plane), and the accumulators take them in candidate
// order: acc = fma(vA, wA, acc), then fma(vB, wB, acc) - bit for bit OpsF32's chain (a missing second candidate adds
// fma(0, w, acc) = acc). Its rows (<= 32 words) are staged two per load instruction with addresses formed on the
// vector ALU (stage_round_v): the eight scalar index loads and 64-bit scalar address computations per wave were
// half of the scalar instructions.
template <int CT_, bool GAUSS, bool RUNS_ = false>
struct OpsPair {
    static constexpr int CT = CT_;
    static constexpr bool RUNS = RUNS_; // carries the run-wise write-out (store_runs): the per-molecule kernel and voxelize_pair_runs_kernel
    static constexpr bool GROUPED = false;
    static constexpr bool VSTAGE = true;
    static constexpr bool PRESTAGE = false;
    // no per-wave sphere / box cull of the staged rows (reaches_subtile): ~35 vector instructions per wave and round to drop
    // 1-2 of a wave's ~8 candidates at 17 (Gaussian) or 12 (binary) instructions each - same box, culled -> not culled,
    // kernel: forward_single 0.121 -> 0.118 ms, 8 types 0.173 -> 0.167, cfg-3 x 256 0.238 -> 0.217 (profiles/r04_narrow.txt)
    static constexpr bool CULL = false;
    typedef float2v Acc[(CT + 1) / 2];
    static constexpr int WORDS = 1;
    static constexpr int WW = CT < 4 ? 4 : CT;       // weight words staged per row (prep pads rows of fewer than 4 channels)
    static constexpr int SW = cand_stride_words(CT); // row stride in LDS, words
    static_assert(16 + WW <= 32, "two rows per load instruction");
    static __device__ __forceinline__ void zero(Acc &acc) {
#pragma unroll
        for (int c = 0; c < (CT + 1) / 2; ++c) acc[c] = (float2v){0.0f, 0.0f};
    }
    static __device__ __forceinline__ LaneCtx ctx(int lane, int wave, int x0, int y0, int z0, int zt_lo, int cbase, const VoxParams &P) {
        const int lz = lane & (SUBZ - 1), ly = (lane >> SUBZ_SH) & (SUBY - 1);
        LaneCtx L;
        L.ix = x0; // (x0 is a multiple of SUBX: every lane evaluates the x0 and the x0 + 1 plane)
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
    static __device__ __forceinline__ void tables(LaneCtx &, char *, const VoxParams &, int) {}
    static __device__ __forceinline__ void walk(Acc &acc, unsigned long long mask, const unsigned *un, int lane, const LaneCtx &L,
                                                const VoxParams &, const double *__restrict__, const float *__restrict__) {
        const bool upper = lane >= 32; // this lane evaluates the pair's second candidate
        while (mask) {
            const int s0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            const bool two = mask != 0; // (uniform)
            int s1 = s0;
            if (two) {
                s1 = __builtin_ctzll(mask);
                mask &= mask - 1;
            }
            const bool valid = !upper || two;
            const unsigned *r = un + (upper ? s1 : s0) * SW;
            const double2 Pxy = *reinterpret_cast<const double2 *>(r);     // px, py
            const double2 PzT = *reinterpret_cast<const double2 *>(r + 4); // pz, T
            const double dx0 = Pxy.x - L.gx, dx1 = Pxy.x - L.gx1, dy = Pxy.y - L.gy, dz = PzT.x - L.gz;
            const double dy2 = dy * dy, dz2 = dz * dz;
            const double d2a = (dx0 * dx0 + dy2) + dz2; // cdist order, no fma
            const double d2b = (dx1 * dx1 + dy2) + dz2;
            const bool hita = valid && d2a <= PzT.y, hitb = valid && d2b <= PzT.y;
            float va, vb;
            if (GAUSS) {
                const float k = __uint_as_float(r[8]);
                va = hita ? __builtin_amdgcn_exp2f(k * (float)d2a) : 0.0f;
                vb = hitb ? __builtin_amdgcn_exp2f(k * (float)d2b) : 0.0f;
            } else {
                va = hita ? 1.0f : 0.0f;
                vb = hitb ? 1.0f : 0.0f;
            }
            // lanes 32-63 of the first register <-> lanes 0-31 of the second: afterwards v0 / v1 hold the first / second
            // candidate's value at the voxel THIS lane accumulates (x0 plane in lanes 0-31, x0 + 1 plane in lanes 32-63)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            const float v0 = __uint_as_float(sw[0]), v1 = __uint_as_float(sw[1]);
            const float *f0 = reinterpret_cast<const float *>(un + s0 * SW + 16), *f1 = reinterpret_cast<const float *>(un + s1 * SW + 16);
            if constexpr (CT == 1) {
                acc[0].x = fmaf(v0, f0[0], acc[0].x);
                acc[0].x = fmaf(v1, f1[0], acc[0].x);
            } else {
                const float2v p0 = (float2v){v0, v0}, p1 = (float2v){v1, v1};
#pragma unroll
                for (int c = 0; c < CT / 2; ++c) acc[c] = __builtin_elementwise_fma(p0, *reinterpret_cast<const float2v *>(f0 + 2 * c), acc[c]);
#pragma unroll
                for (int c = 0; c < CT / 2; ++c) acc[c] = __builtin_elementwise_fma(p1, *reinterpret_cast<const float2v *>(f1 + 2 * c), acc[c]);
            }
        }
    }
    static __device__ __forceinline__ void write(const Acc &acc, bool any, unsigned *un, int tid, int lane, int wave, int NW,
                                                 int b, const LaneCtx &L, int x0, int y0, int z0, void *out, const VoxParams &P) {
        write_slab<CT, RUNS>(acc, any, reinterpret_cast<float *>(un), tid, lane, wave, NW, b, L.cbase, x0, y0, z0,
                             static_cast<float *>(out), P);
    }
};

} // namespace mvx
