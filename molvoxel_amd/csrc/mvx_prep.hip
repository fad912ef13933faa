// mvx_prep.hip - the pre-pass of the binned pipeline (gfx950): per-atom records and the ordered candidate lists.
//
//   chan_aux_kernel  channel-wise radii for features: max radius, and per chunk of 32 channels the distinct radii ("slots")
//   prep_kernel      one thread per atom: rigid transform in the reference's fp64 op order, exact box cull +
//                    per-axis reference-block cull folded into an admitted voxel-index range, exact membership
//                    threshold T on d2, gaussian coefficient k -> one 64-B record per atom (+ packed channel weights).
//   xbin_kernel      ordered (ballot/prefix, no atomics) candidate lists: per (molecule, x-slab) and, from
//                    those, one 512-B candidate line per output slab.
//   transform_kernel do_transform for T / RandomTransform objects on device tensors.
#include "mvx_device.h"

namespace mvx {

LaunchEvents &launch_events() {
    static thread_local LaunchEvents ev;
    return ev;
}
void set_launch_events(hipEvent_t start, hipEvent_t stop) {
    launch_events().start = start;
    launch_events().stop = stop;
}
bool launch_events_pending() { return launch_events().start != nullptr; }

#ifdef MVX_DIAG
__device__ unsigned long long *g_diag_xb = nullptr; // xbin_kernel: 8 x 8 B per block, stamps by thread 0
#define XB_STAMP(i) do { if (g_diag_xb && threadIdx.x == 0) g_diag_xb[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
hipError_t set_diag_buffer_xb(void *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_xb), &p, sizeof(p)); }
#else
#define XB_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// channel-wise radii for features: max radius (float32) and, per chunk of 32 channels, the distinct radii
// ------------------------------------------------------------------------------------------------
// numpy/voxelizer.py:213-224 evaluates one membership test and one density per channel; channels that share a radius share
// both. One wave per chunk of 32 channels (the chunk a grouped voxelize workgroup owns): lane l < 32 holds channel 32 k + l.
// A "slot" is one distinct radius of the chunk; slots are numbered by DESCENDING radius (float32 bit patterns of valid radii
// order like their values; invalid radii - non-positive, non-finite: threshold -1, never a hit - share key 0, the last
// slot), so the walk can stop at the first slot no lane hits. A chunk has at most 32 channels, hence at most 32 slots: no
// channel count and no set of radii needs another kernel.
__global__ void __launch_bounds__(256) chan_aux_kernel(const float *radii, int C, int density, float sigma32, float *rmax,
                                                       ChanGroups *groups, int *chan_slot) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int nchunk = (C + 31) / 32;
    for (int chunk = wv; chunk < nchunk; chunk += nwv) { // (wave-uniform)
        const int c = 32 * chunk + (lane & 31);
        const bool live = lane < 32 && c < C;
        const float r = live ? radii[c] : 0.0f;
        const unsigned key = (live && r > 0.0f && r < 3.0e38f) ? __float_as_uint(r) : 0u;
        bool first = live; // no earlier channel of the chunk has this key
        for (int j = 0; j < 32; ++j) {
            const unsigned kj = (unsigned)__shfl((int)key, j, 64);
            const bool lj = __shfl((int)live, j, 64) != 0;
            if (j < lane && lj && kj == key) first = false;
        }
        int slot = 0; // distinct keys of the chunk above this one
        for (int j = 0; j < 32; ++j) {
            const unsigned kj = (unsigned)__shfl((int)key, j, 64);
            const bool fj = __shfl((int)first, j, 64) != 0;
            if (fj && kj > key) ++slot;
        }
        const int nslots = __popcll(__ballot(first));
        ChanGroups &G = groups[chunk];
        if (live) chan_slot[c] = slot;
        if (first) {
            G.slot[slot].T = d2_threshold(r);
            G.slot[slot].k = density == MVX_GAUSSIAN ? gauss_coeff(r, sigma32) : 0.0f;
            G.slot[slot].pad = 0;
        }
        if (lane == 0) {
            G.nslots = nslots;
            G.pad[0] = G.pad[1] = G.pad[2] = 0;
        }
    }
    if (threadIdx.x == 0) { // the culls' radius: max(radii) as numpy evaluates it (numpy/voxelizer.py:138)
        float m = radii[0];
        for (int c = 1; c < C; ++c) m = radii[c] > m ? radii[c] : m;
        rmax[0] = m;
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

hipError_t launch_chan_aux(const float *radii, int32_t C, int32_t density, float sigma32, float *rmax, ChanGroups *groups,
                           int32_t *chan_slot, hipStream_t s) {
    hipLaunchKernelGGL(chan_aux_kernel, dim3(1), dim3(256), 0, s, radii, C, density, sigma32, rmax, groups, chan_slot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// prep: per-atom records
// ------------------------------------------------------------------------------------------------
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
    xbin_kernel(const uint2 *__restrict__ xp, const int64_t *__restrict__ offsets, int64_t n_one, int b0, int nsx, unsigned nsx_inv, int nsy, int nzc,
                int NW, uint2 *__restrict__ xlist, uint2 *__restrict__ slist, uint2 *__restrict__ slist_ext) {
    __shared__ uint2 xs[XLN]; // (one-wave blocks serve molecules of <= 256 atoms: XLN = 256)
    constexpr int NWV = THREADS / 64; // waves per block
    __shared__ int wcnt[2][NWV];
    __shared__ int any_overflow;
    __shared__ uint2 line[NWV][NQ * SLOTS]; // the NQ slab lines each wave is building
    // (blockIdx.x = (molecule - b0) * nsx + sx, split without the run-time integer division: nsx_inv = ceil(2^32 / nsx))
    const unsigned bq = nsx == 1 ? blockIdx.x : __umulhi(blockIdx.x, nsx_inv);
    const int b = b0 + (int)bq, sx = (int)(blockIdx.x - bq * (unsigned)nsx);
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
            const int sq = nzc == 1 ? sl : sl / nzc; // (one slab per row - every grid of up to 64 voxels: no division)
            sy[q] = (sl < nslab) ? sq : 255; // 255: beyond the grid, matches no entry
            zt_lo[q] = (sl - sq * nzc) * NW;
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
    const unsigned nsx_inv = nsx > 1 ? (unsigned)((0x100000000ull + (unsigned long long)nsx - 1) / (unsigned long long)nsx) : 0u;
    if (max_atoms <= 256) { // small molecules (one round of pass A for a single wave): one-wave blocks
        int parts = 1;
        while (parts * 4 < nslab && parts < 4 && (long long)nb * nsx * parts < 8192) parts *= 2;
        hipLaunchKernelGGL((xbin_kernel<64, 256, 4, 4>), dim3((unsigned)(nb * nsx), (unsigned)parts), dim3(64), 0, s, xp, offsets, n_one, b0, nsx,
                           nsx_inv, nsy, nzc, NW, xlist, slist, slist_ext);
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
            hipLaunchKernelGGL((xbin_kernel<1024, 2 * XL_LDS, CHUNKS, 4>), grid, dim3(1024), 0, s, xp, offsets, n_one, b0, nsx, nsx_inv, nsy, nzc, \
                               NW, xlist, slist, slist_ext);                                                                          \
        else                                                                                                                          \
            hipLaunchKernelGGL((xbin_kernel<1024, 4 * XL_LDS, CHUNKS, 1>), grid, dim3(1024), 0, s, xp, offsets, n_one, b0, nsx, nsx_inv, nsy, nzc, \
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
    hipLaunchKernelGGL((xbin_kernel<256, XL_LDS, CHUNKS, 4>), grid, dim3(256), 0, s, xp, offsets, n_one, b0, nsx, nsx_inv, nsy, nzc, NW, xlist, \
                       slist, slist_ext)
    if (max_atoms <= 1024) MVX_XBIN_256(4);
    else if (max_atoms <= 2048) MVX_XBIN_256(8);
    else MVX_XBIN_256(16);
#undef MVX_XBIN_256
    return hipGetLastError();
}

} // namespace mvx
