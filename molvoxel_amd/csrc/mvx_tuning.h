// mvx_tuning.h - every measured constant of the kernels and of the host-side plan, in one place, each with the measurement
// that chose it (MI355X, same-box A/B runs; the tables are in profiles/, the history in profiles/HISTORY.md).
#pragma once

namespace mvx {

// ---- write-out rounds (channels transposed through LDS per round) --------------------------------------------------------
// vector-ALU slab kernels: 512 threads read back exactly one 32-row tile per round. cfg-2 x 256, of peak: 16 channels per
// round 0.783-0.787, 8: 0.790, 4: 0.792-0.795 - smaller store bursts interleave better between a unit's workgroups.
constexpr int CR_F32 = 4;
constexpr int MX_CR = 8;      // matrix-core path: D holds 4 channels x 2 planes per lane and round
constexpr int CR64 = 8;       // float64 grids

// ---- pacing of the store stream (profiles/r03_round_pacing.txt) -----------------------------------------------------------
// s_sleep counts units of 64 cycles. An empty slab holds its zero fill back EMPTY_HOLD units (ligand batches 6.2 -> 6.7 TB/s;
// 16 / 32 / 48 / 64 / 80 / 100 units: +0.9 / 2.8 / 4.7 / 7.5 / 7.0 / 3.7 %) and sends it in pieces of two store instructions
// EMPTY_SPLIT units apart (6.55 -> 6.83 TB/s; 8 / 16 / 24 / 32 units: +2 / +3.5 / +4.3 / +3.6 %).
constexpr int EMPTY_HOLD = 64;
constexpr int EMPTY_SPLIT = 20;
// A light slab of a store-bound launch waits after each write-out round about as long as a compute unit needs to drain the
// round's bytes: 8 channels x 64 NW voxels x 4 B at ~12.8 B per cycle = 2.5 NW units (NW = 8: 20 units = 1280 cycles; cfg-2 x
// 256 of peak, 4 / 16 / 20 / 24 / 30 / 40 units: +0.5 / +2.3 / +2.5...4.7 / +2 / +0.5 / -7 %). Issued as NW/2 sleeps of
// ROUND_SLEEP_STEP units (s_sleep takes an immediate). Re-measured after the staging lost its scalar work (the slabs reach
// their write-out sooner; profiles/r04_staging.txt), same box, of peak: 4 / 5 / 6 / 7 / 8 / 10 units per step 0.798 / 0.811-0.829 /
// 0.818-0.833 / 0.826 / 0.809 / 0.76: 6 is +0.4 ... 0.8 % over 5 in each of three runs.
constexpr int ROUND_SLEEP_STEP = 6;
// ... only slabs of at most this many candidates: heavier slabs are bound by their walk and lose by waiting (1.5 A radius:
// -5 % with every slab paced, +1 % with the limit; 2.0 A: +2 %)
constexpr int PACE_MAX_CANDIDATES = 48;
// ... and only in launches of at least PACE_ROUNDS_MIN_WGS workgroups (96 cfg-2 molecules; unpaced -> paced, of peak: 16
// molecules 0.734 -> 0.715, 32: 0.767 -> 0.745, 64: 0.773 -> 0.785, 128: 0.775 -> 0.801, 256: 0.768 -> 0.801); empty slabs
// are held back in launches of more than PACE_EMPTY_MIN_WGS workgroups (a small launch would just start later)
// (after the staging change, same box, kernel of peak unpaced / paced: 16 molecules 0.757 / 0.735, 32: 0.77 / 0.77, 64 in a
// sustained loop 0.791 / 0.807 - but 64 in bursts of 20 calls, tools/variants.py: 0.394 / 0.417 ms per call: the limit stays)
constexpr long long PACE_ROUNDS_MIN_WGS = 49152;
constexpr long long PACE_EMPTY_MIN_WGS = 4096;

// ---- occupancy targets ------------------------------------------------------------------------------------------------------
// waves per SIMD the 1024-thread slab variants (whole rows of 65 ... 128 voxels: 9 ... 16 waves) are compiled for: 8 = 64
// registers, two or three workgroups per unit (D = 72: 4.16 TB/s against 3.75 with 4 = 128 registers, one workgroup)
constexpr int BIG_WAVES_PER_SIMD = 8;

// ---- host-side plan (plan_call in mvx_plan.hip) ---------------------------------------------------------------------------
// bytes of pre-pass data (records, keys, feature rows, slab lines: re-read ~20 times) per voxelize launch: what stays in
// the 256 MiB Infinity Cache. One launch over 512 cfg-2 molecules (544 MB) ran at 0.654 of peak, in two chunks at 0.755.
constexpr double MALL_BUDGET = 288.0e6;
// One launch for the whole call (voxelize_pair_kernel: two slabs per workgroup, one workgroup per compute unit at a time)
// instead of prep -> xbin -> voxelize. us per call binned / one launch (tools/route_sweep.py, profiles/r04_route_sweep.txt);
// "workgroups" below counts slabs as plan_call does (a pair kernel workgroup serves two). One molecule on a 64^3 grid, C = 32
// (512 slabs = one round of 256 workgroups): 50 atoms 16.1 / 9.5, 4 000 19.5 / 15.0, 8 000 26.4 / 19.0, 16 000 36.5 / 28.5,
// 24 000 48.7 / 38.2, 32 000 173 / 47.5, 48 000 (two segments) 253 / 76; C = 4: 4 000 17.8 / 10.6; 48^3: 16 000 141 / 36.6.
// Several rounds of workgroups repeat the front (scan, stage, walk) while the binned pipeline shares its pre-pass:
// 1 024 slabs (two pockets, or C = 64): 500 atoms each 21.0 / 20.4, 2 000 23.1 / 23.4, 4 000 25.4 / 26.7; 1 536 slabs: 27.5 / 28.8
// ... 31.1 / 38.0; ligands (50 atoms, C = 16): 2 per call 15.7 / 10.7, 4 per call 20.3 / 17.9, 8 per call 30.7 / 31.9, 16: 51 / 83;
// two cfg-3 molecules (576 slabs) 19.8 / 14.5, four 17.9 / 18.0.
constexpr long long DIRECT_MAX_WORKGROUPS = 2048;
// ... and at most this many atom tests (slabs x atoms of their molecule). One round of workgroups (up to 512 slabs): never
// the limit in practice (131 072 atoms at 512 slabs)
constexpr long long DIRECT_MAX_ATOM_TESTS = 64ll << 20;
// ... two rounds (513 ... 1 024 slabs: two molecules per call)
constexpr long long DIRECT_MAX_ATOM_TESTS_TWO = 600000;
// ... three or four rounds (ligand-sized molecules only)
constexpr long long DIRECT_MAX_ATOM_TESTS_MANY = 300000;
// ... several channel chunks (C > 32) scan, stage and walk once per chunk: one molecule, C = 64, binned / one launch: D = 64
// 500 atoms 20.6 / 20.3, 2 000 22.0 / 23.5, 4 000 24.2 / 26.6; D = 48 1 700 atoms 18.2 / 19.4 (0.98 M tests); D = 32 500 atoms
// 14.0 / 9.3
constexpr long long DIRECT_MAX_ATOM_TESTS_CHUNKED = 600000;
// float64 grids of more channels than this take the matrix-core slab kernel (8 or 16 channels padded to a chunk of 32 there:
// 1.7 / 2.9 TB/s against 2.2 / 3.2 for the general loop)
constexpr int MX64_MIN_C = 16;

} // namespace mvx
