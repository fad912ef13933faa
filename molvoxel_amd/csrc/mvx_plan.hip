// mvx_plan.hip - plan_call(): the decision table of a call as a pure host function (no device code in this TU).
// Every number here is a measurement; the constants and what chose them are in mvx_tuning.h.
#include "mvx_plan.h"

#include <algorithm>
#include <cmath>

#include "mvx_internal.h"

namespace mvx {

namespace {

int pick_ct(int C) {
    if (C <= 1) return 1;
    if (C <= 4) return 4;
    if (C <= 8) return 8;
    if (C <= 16) return 16;
    return 32;
}

// slab = SUBX x SUBY x (SUBZ*NW) voxels, NW waves side by side along z. Whole rows (NW = row length in sub-tiles) up to
// `max_waves`; longer rows are cut into chunks of 8 sub-tiles (256-B runs).
// whole_rows (the binned float32 pipeline): rows of 65 ... 128 voxels stay whole too (9 ... 16 waves, the 1024-thread
// kernel variants, two or three workgroups per compute unit) unless D % 32 == 0. A row cut at 256 B leaves pieces that
// share 64-byte blocks with their neighbours whenever rows are not multiples of 64 B - 16 molecules per call, C = 32,
// TB/s of grid bytes, chunks -> whole rows: D = 72 1.80 -> 4.16, 88 1.94 -> 4.02, 104 2.12 -> 4.03, 120 2.25 -> 3.81,
// 80 3.70 -> 4.12, 112 3.95 -> 4.02, 65 2.13 -> 2.57; D = 96 4.26 -> 4.28 and D = 128 5.13 -> 4.22 keep their chunks
// (tools/odd_d_probe.py ROW_SWEEP=1, profiles/r03_odd_dimensions.txt).
void plan_slabs(int D, int max_waves, bool whole_rows, int force_nw, mvx_plan &p, bool quad_rows = false) {
    p.nsx = (D + SUBX - 1) / SUBX;
    p.nsy = (D + SUBY - 1) / SUBY;
    const int nsz = (D + SUBZ - 1) / SUBZ;
    p.nw = nsz <= max_waves ? nsz : max_waves;
    if (whole_rows && nsz > max_waves && nsz <= 16 && D % 32 != 0) p.nw = nsz;
    // longer rows that are not multiples of 64 B: as few, equally long chunks as 16 waves allow (every cut shares a 64-byte
    // block between two workgroups) - D = 136 1.90 -> 2.10 TB/s, 152 2.28 -> 2.50, 168 2.01 -> 2.19, 200 (C = 16) 1.68 ->
    // 1.89; multiples of 64 B keep chunks of 8 (D = 144 3.36 against 2.54-2.67, 160 3.60 against 2.81-3.15)
    if (whole_rows && nsz > 16 && D % 16 != 0) {
        const int k = (nsz + 15) / 16;
        p.nw = (nsz + k - 1) / k;
    }
    // rows of 8 k + 4 sub-tiles cut into chunks of 8 (D = 96, 160, 224: multiples of 128 B) leave a last chunk whose slabs
    // run half empty - four of eight waves idle through every barrier. The store-bound 32-channel launches do not notice (D = 96,
    // kernel TB/s of grid bytes, chunks of 8 / 4: C = 32 5.72 / 5.66, C = 64 5.60 / 5.33); narrow chunks, which are bound by their
    // walk, gain what the idle waves cost: C = 16 4.39 -> 5.91 (4 000 atoms) and 3.73 -> 4.67 (13 500), C = 8 3.83 -> 4.63 and
    // 3.11 -> 3.62, C = 4 3.03 -> 3.32 (tools/d_kernel_probe.py, 19 molecules per call)
    if (whole_rows && p.ct < 32 && nsz > 8 && nsz % 8 == 4 && p.nw == 8) p.nw = 4;
    // One or four channels per workgroup (forward_single, a few element types): the multi-sub-tile kernel (two or four sub-tiles
    // per wave) needs an even number of waves per slab, and these launches store too little to care where a row is cut. Rows
    // of 9 ... 15 sub-tiles in chunks of eight instead of whole (grids of whole 16-byte quads only; kernel ms whole / 4 / 8,
    // cfg-2 density, tools/jobs/narrow_bigd_sweep.sh): C = 1 D = 88 0.155 / 0.097 / 0.108, 104 0.184 / 0.100 / 0.096, 112 0.124 /
    // 0.098 / 0.094, 120 0.190 / 0.097 / 0.091; C = 4 D = 88 0.186 / 0.126 / 0.124, 104 0.200 / 0.168 / 0.115, 112 0.147 / 0.125 /
    // 0.114, 120 0.213 / 0.170 / 0.110 (whole calls 0.27 -> 0.19 ms at D = 120). Nine sub-tiles (D = 72) take chunks of four (0.141
    // / 0.107 / 0.120 and 0.164 / 0.130 / 0.137); ten (D = 80) stay whole - five waves of two sub-tiles, 0.111 / 0.099 / 0.114 with
    // the smaller pre-pass of the whole row. Eight channels keep whole rows (D = 88 0.228 / 0.309 / 0.254).
    if (whole_rows && quad_rows && p.ct <= 4 && nsz > 8 && nsz < 16 && p.nw == nsz && nsz != 10) p.nw = nsz == 9 ? 4 : 8;
    // ... and forward_single (one channel) on rows of five or seven sub-tiles (D = 40, 56): chunks of four - kernel 0.128 -> 0.116 and
    // 0.126 -> 0.099 ms, calls 0.162 -> 0.156 and 0.157 -> 0.141 (four channels: no gain; three sub-tiles, D = 24: whole rows win)
    if (whole_rows && quad_rows && p.ct == 1 && (nsz == 5 || nsz == 7) && p.nw == nsz) p.nw = 4;
    if (force_nw > 0 && force_nw <= 16) p.nw = std::min(force_nw, nsz); // "nw" measurement knob
    p.nzc = (nsz + p.nw - 1) / p.nw;
}

} // namespace

mvx_plan plan_call(const mvx_plan_query &q, const PlanKnobs &k) {
    mvx_plan p{};
    const int D = q.dimension, C = q.C, B = q.B;
    const bool f64 = q.precision == 64;
    const bool chanwise = q.radii_type == MVX_RADII_CHANNEL && q.mode == MODE_FEATURES;
    const int bd = q.blockdim > 0 ? q.blockdim : 8;
    const int nb = (D + bd - 1) / bd;

    // ---- channels per workgroup (register accumulators per lane); more channels -> several channel chunks -----------
    // float64 grids: more than MX64_MIN_C channels with scalar / atom-wise radii take chunks of 32 on 8-wave slabs through
    // the matrix-core slab kernel (128 registers, two workgroups per unit); everything else 16 per workgroup in the general
    // slab loop (105 VGPRs, two 8-wave workgroups per unit).
    const bool mx64 = f64 && C > MX64_MIN_C && !chanwise && k.max_ct64 >= 32 && k.max_ct >= 32 && k.force_nw == 0;
    p.ct = mx64 ? 32 : pick_ct(std::min(C, f64 ? std::min(k.max_ct, 16) : k.max_ct));
    p.ncc = (C + p.ct - 1) / p.ct;
    p.grouped = (chanwise && !f64) ? 1 : 0;
    if (p.grouped) { // chunks of 32 channels on the matrix-core path, whatever C is
        p.ct = 32;
        p.ncc = (C + 31) / 32;
    }

    // ---- route -------------------------------------------------------------------------------------------------------
    plan_slabs(D, 8, false, k.force_nw, p);
    // One launch for the whole call (voxelize_pair_kernel) when
    // the per-workgroup atom scan is cheap next to the slab's stores: per-molecule forward() calls, and a few small
    // molecules per call. Bigger jobs amortise the binning pre-pass.
    // Whole-row slabs only (D <= 64: with rows cut in two the one-launch route loses at every size measured - D = 68 / 72 /
    // 76, us per call one launch / binned: 4000-atom-density pocket 57 / 23, 65 / 24, 69 / 27; 8 atoms 24 / 18, 26 / 19,
    // 28 / 23). Channel-wise radii for features always take the grouped launch of the binned pipeline.
    bool direct = false;
    if (!f64 && !chanwise && p.nw <= 8 && p.nzc == 1 && (long long)B * p.ncc <= 65535) {
        if (k.direct_mode >= 0) direct = k.direct_mode == 1;
        else {
            const long long per_mol = (long long)p.nsx * p.nsy * p.nzc;
            const long long wgs = (long long)B * p.ncc * per_mol;
            const long long limit = p.ncc > 1 ? DIRECT_MAX_ATOM_TESTS_CHUNKED
                                              : (wgs <= 512 ? DIRECT_MAX_ATOM_TESTS : (wgs <= 1024 ? DIRECT_MAX_ATOM_TESTS_TWO : DIRECT_MAX_ATOM_TESTS_MANY));
            direct = wgs <= DIRECT_MAX_WORKGROUPS && (long long)p.ncc * per_mol * q.total_atoms <= limit;
        }
    }
    p.route = f64 ? (mx64 ? MVX_ROUTE_F64_MX : MVX_ROUTE_F64_DENSE) : (direct ? MVX_ROUTE_DIRECT : MVX_ROUTE_BINNED);
    if (p.route == MVX_ROUTE_BINNED) plan_slabs(D, 8, true, k.force_nw, p, !f64 && D % 4 == 0 && q.out_aligned16 != 0);
    const long long per_mol = (long long)p.nsx * p.nsy * p.nzc;

    // ---- remainder channels --------------------------------------------------------------------------------------------
    // Channel counts that are not a multiple of the chunk width (C = 33 ... 63, 65 ...): the binned float32 pipeline runs
    // the full chunks with the wide kernel and the remainder with the narrowest kernel that holds it (C = 33: 32 + 1,
    // C = 40: 32 + 8) - a second, small voxelize launch over the same candidate lines - instead of a whole extra chunk of
    // `ct` accumulators that are mostly padding (C = 33 cost +49 % for +3 % of the bytes).
    p.nfull = p.ncc;
    p.ct_rem = 0;
    if (p.route == MVX_ROUTE_BINNED && !p.grouped && p.ncc > 1 && C % p.ct != 0) {
        const int rem = pick_ct(C - (C / p.ct) * p.ct);
        if (rem < p.ct) { // (a remainder of more than half a chunk needs the wide kernel anyway: one launch)
            p.ct_rem = rem;
            p.nfull = C / p.ct;
        }
    }
    // channel weights per atom, zero padded; feature rows that already are that wide are read in place, anything else
    // (one-hot types, 1, padding) is packed by prep. Grouped launches read the caller's rows in place whatever C is.
    p.cpad = p.ct_rem ? p.nfull * p.ct + (p.ct_rem < 4 ? 4 : p.ct_rem) : ((p.ncc > 1) ? p.ncc * p.ct : (p.ct < 4 ? 4 : p.ct));
    p.weights_in_place = (q.mode == MODE_FEATURES && (C == p.cpad || p.grouped)) ? 1 : 0;

    // ---- molecule chunks -------------------------------------------------------------------------------------------------
    // gridDim.y limit; and a chunk's pre-pass output and inputs (records, binning keys, feature rows, slab lines) are
    // re-read ~20 times by its voxelize launch: while they fit the Infinity Cache the row loads are served on-die, so
    // larger batches are cut to MALL_BUDGET and pre-pass / voxelize launches alternate chunk by chunk.
    const int max_mol = 65535 / p.ncc;
    p.nchunk = (B + max_mol - 1) / max_mol;
    if (!f64 && B > 0) {
        const double per_atom = 64.0 + 8.0 + 4.0 * (double)((C + 3) / 4 * 4);
        const double ws = (double)q.total_atoms * per_atom + (double)B * (double)per_mol * 512.0;
        const int mall_chunks = (int)std::min<double>(std::ceil(ws / k.mall_budget), (double)std::max(1, B));
        p.nchunk = std::max(p.nchunk, mall_chunks);
    }
    if (k.pipeline > 1 && B >= 4 * k.pipeline) p.nchunk = std::max(p.nchunk, k.pipeline);
    if (p.nchunk < 1) p.nchunk = 1;

    // ---- write-out path ----------------------------------------------------------------------------------------------------
    // 16-B stores need whole float4 groups per row (D % 4 == 0) and a 16-B aligned grid; anything else (odd dimensions, a
    // slice `grid[i]` of a batch grid whose slices are not 16-B multiples) is written run by run (store_runs)
    p.vec_store = (D % (f64 ? 2 : 4) == 0 && q.out_aligned16) ? 1 : 0;
    p.xcd_ranges = (!f64 && !p.vec_store && p.nzc == 1) ? 1 : 0;
    // a sub-tile lies inside one reference block when its edges divide blockdim (or there is a single block): the block
    // cull is then wave-uniform and already folded into the candidate ranges; otherwise every lane checks its voxel's index
    p.lane_range = !(nb == 1 || (bd % SUBX == 0 && bd % SUBY == 0 && bd % SUBZ == 0)) ? 1 : 0;

    // ---- pacing (mvx_tuning.h) ---------------------------------------------------------------------------------------------
    const long long wgs = (long long)B * per_mol * p.ncc;
    p.pace = wgs >= PACE_ROUNDS_MIN_WGS ? 2 : (wgs > PACE_EMPTY_MIN_WGS ? 1 : 0);
    return p;
}

} // namespace mvx

extern "C" int mvx_plan_call(const mvx_plan_query *query, mvx_plan *plan) {
    if (!query || !plan) return MVX_ERR_INVALID;
    if (query->dimension < 1 || query->dimension > 1020 || query->B < 0 || query->C <= 0 || query->mode < 0 || query->mode > 2 ||
        query->radii_type < MVX_RADII_SCALAR || query->radii_type > MVX_RADII_CHANNEL)
        return MVX_ERR_INVALID;
    *plan = mvx::plan_call(*query, mvx::PlanKnobs{});
    return MVX_OK;
}
