// mvx_f64.hip - float64 grids (precision = 64; gfx950): the reference then keeps distances, ratios, densities and sums in
// float64 (numpy/voxelizer.py:33-34, 544-560).
//
//   voxelize64_kernel      chunks of 32 channels, scalar / atom-wise radii: the slab body (mvx_slab_body.inc) with OpsMx64 -
//                          v_mfma_f64_16x16x4_f64, four candidates per instruction, bit for bit the fma chain in candidate order.
//   voxelize_dense_kernel  the general slab loop (<= 16 channels, or channel-wise radii): grid-stride over all slabs. Per
//                          slab: rounds of 64 entries over the primary + extension line (<= 255 candidates), or, beyond
//                          that, wave 0 compacts the (molecule, x-slab) list in rounds of LCAP entries into an LDS list and
//                          the rows are staged in rounds of dcap; same walk and write-out.
//   LDS map of the dense kernel (dynamic, 16-B aligned): int list[LCAP] | uint32 zr[LCAP] | int nlist |
//                          union { dcap rows ; tile }, LCAP = 64 * min(NW, 4).
#include "mvx_device.h"

namespace mvx {

__host__ __device__ __forceinline__ int row_stride_doubles(int NW) { return SUBZ * NW + 8; }
// general slab loop: rows of 16 + 2*ct words, 64 per round (p.dcap = 64), or the write-out tile
static size_t dense64_lds_bytes(int32_t ct, int32_t NW) {
    const int lcap = 64 * (NW < 4 ? NW : 4);
    const size_t rows = (size_t)64 * (16 + 2 * ct) * 4;
    const size_t tile = (size_t)(ct < CR64 ? ct : CR64) * RPC * row_stride_doubles(NW) * 8;
    return (size_t)8 * lcap + 16 + (rows > tile ? rows : tile);
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
static size_t voxelize_mx64_lds_bytes(int32_t NW) {
    const size_t rows = (size_t)64 * MX64_SW * 4;
    const size_t tile = (size_t)CR64 * RPC * row_stride_doubles(NW) * 8;
    return rows > tile ? rows : tile;
}

template <bool GAUSS, bool LANE_RANGE>
struct OpsMx64 {
    static constexpr bool RUNS = false;
    static constexpr int CT = 32;
    static constexpr bool GROUPED = false;
    static constexpr bool VSTAGE = false;
    static constexpr bool CULL = true;
    static constexpr bool PRESTAGE = false;
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
    // the 2^(j/64) table of exp_nonpos64, behind the rows / tile region (P.dcap bytes)
    static __device__ __forceinline__ void tables(LaneCtx &L, char *smem, const VoxParams &P, int tid) {
        double *tab = reinterpret_cast<double *>(smem + P.dcap);
        if (tid < 64) tab[tid] = EXP2_64TH[tid];
        L.gtab = tab;
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
    static constexpr bool PRESTAGE = false;
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

// float64 grids, chunks of 32 channels, scalar / atom-wise radii: the slab body with OpsMx64 (128 registers)
template <bool GAUSS, bool LANE_RANGE, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 512 ? 4 : 2))
    voxelize64_kernel(const unsigned *__restrict__ rec, const unsigned *__restrict__ w, const uint2 *__restrict__ slist,
                      const uint2 *__restrict__ slist_ext, double *__restrict__ out, const VoxParams P) {
    typedef OpsMx64<GAUSS, LANE_RANGE> Ops;
    constexpr int CT = 32;
    constexpr bool GROUPED = false;
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
// dispatch
// ------------------------------------------------------------------------------------------------
struct KernelKey64 {
    int ct;
    bool gauss, chanwise, lane_range;
};

// Calls fn.template operator()<CT, GAUSS, CHANWISE, LANE_RANGE>() for the instantiation `k` names.
template <typename Fn>
static hipError_t for_kernel64(const KernelKey64 &k, Fn &&fn) {
#define MVX_CASE(CT_, G_, CW_, LR_) \
    if (k.ct == CT_ && k.gauss == G_ && k.chanwise == CW_ && k.lane_range == LR_) return fn.template operator()<CT_, G_, CW_, LR_>();
#define MVX_CASES_CT(CT_)             \
    MVX_CASE(CT_, true, false, false)  \
    MVX_CASE(CT_, false, false, false) \
    MVX_CASE(CT_, true, false, true)   \
    MVX_CASE(CT_, false, false, true)  \
    MVX_CASE(CT_, true, true, true)    \
    MVX_CASE(CT_, false, true, true)
    MVX_CASES_CT(1)
    MVX_CASES_CT(4)
    MVX_CASES_CT(8)
    MVX_CASES_CT(16)
    MVX_CASES_CT(32)
#undef MVX_CASES_CT
#undef MVX_CASE
    return hipErrorInvalidValue;
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

struct Dense64Fn {
    const VoxArgs &a;
    hipStream_t s;
    template <int CT, bool GAUSS, bool CHANWISE, bool LANE_RANGE>
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
    KernelKey64 k{ct, gauss, chanwise, chanwise ? true : lane_range};
    return for_kernel64(k, Dense64Fn{a, s});
}

} // namespace mvx
