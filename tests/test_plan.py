"""The library's decision table (mvx_plan_call: route, slab plan, channel / molecule chunks, pacing, write-out path) pinned row
by row next to the measurements that chose each row (molvoxel_amd/csrc/mvx_tuning.h, profiles/r04_route_sweep.txt,
profiles/r03_odd_dimensions.txt, profiles/r03_round_pacing.txt). mvx_plan_call is a pure host function: no GPU is needed, nothing is launched.
A change of the rule must come with the measurement that justifies moving a row."""
import pytest

from molvoxel_amd.voxelizer.hip import _lib

BINNED, DIRECT, F64_DENSE, F64_MX = 0, 1, 2, 3


def plan(D, C, B=1, atoms=None, **kw):
    """atoms: per molecule (default: cfg-2 density, 4000 atoms in 31.5^3 A^3 scaled to the grid's volume)"""
    if atoms is None:
        atoms = int(round(4000 * ((D - 1) / 63.0) ** 3))
    return _lib.plan_call(D, C, B, total_atoms=B * atoms, max_atoms=atoms, **kw)


def test_baseline_configurations():
    # cfg-2, the headline: 256 pockets per call, one chunk of 32 channels on whole-row slabs, paced write-out rounds
    p = plan(64, 32, 256, 4000)
    assert (p["route"], p["nsx"], p["nsy"], p["nzc"], p["nw"], p["ct"], p["ncc"], p["nchunk"], p["pace"]) == (BINNED, 32, 16, 1, 8, 32, 1, 1, 2)
    assert p["weights_in_place"] == 1 and p["vec_store"] == 1 and p["lane_range"] == 0 and p["ct_rem"] == 0
    # one cfg-2 pocket per forward() call: 512 slabs = 256 pair workgroups, one launch (15 against 19.5 us binned)
    assert plan(64, 32, 1, 4000)["route"] == DIRECT
    # cfg-1 ligand and cfg-3 (binary types, 48^3, N = 1000) as single calls: one launch
    assert plan(64, 5, 1, 33)["route"] == DIRECT
    assert plan(48, 4, 1, 1000, mode="types")["route"] == DIRECT
    # cfg-5: 128^3, one call = 4096 workgroups -> binned, rows of 512 B cut in two (D % 32 == 0 keeps chunks of 8 waves)
    p = plan(128, 32, 1, 10000, radii_type="atom-wise")
    assert (p["route"], p["nw"], p["nzc"], p["nsx"], p["nsy"], p["pace"]) == (BINNED, 8, 2, 64, 32, 0)
    # cfg-4 x 128 ligands per GPU: 65 536 workgroups of 16 channels, empty slabs and rounds paced
    p = plan(64, 16, 128, 50)
    assert (p["route"], p["ct"], p["pace"]) == (BINNED, 16, 2)


def test_route_rule_rows():
    """profiles/r04_route_sweep.txt (tools/route_sweep.py, voxelize_pair_kernel; us per call binned / one launch) and, for rows cut
    in two, profiles/r03_odd_dimensions.txt."""
    # rows cut in two (D = 65 ... 76) never take the one-launch route: 57 / 23, 65 / 24, 69 / 27 us
    for D in (68, 72, 76):
        assert plan(D, 32)["route"] == BINNED
        assert plan(D, 32, atoms=8)["route"] == BINNED
    # one molecule on up to 512 slabs (one round of pair workgroups): one launch at every size measured - 64^3, C = 32:
    # 8 000 atoms 26.4 / 19.0, 24 000 48.7 / 38.2, 48 000 253 / 76; 48^3, 16 000 atoms 141 / 36.6
    for atoms in (8, 4000, 8000, 24000, 48000):
        assert plan(64, 32, 1, atoms)["route"] == DIRECT
    assert plan(48, 32, 1, 16000)["route"] == DIRECT
    # several channel chunks: one launch up to 0.6 M atom tests - C = 64 at D = 64: 500 atoms 20.6 / 20.3 -> one launch, 2 000
    # atoms 22.0 / 23.5 and the 4 000-atom pocket 24.2 / 26.6 -> binned; D = 24 / 32 dense 14.0 / 9.3 -> one launch
    assert plan(64, 64, 1, 4000)["route"] == BINNED
    assert plan(64, 64, 1, 2000)["route"] == BINNED
    assert plan(64, 64, 1, 500)["route"] == DIRECT
    assert plan(64, 64, 1, 8)["route"] == DIRECT
    assert plan(24, 64)["route"] == DIRECT
    # two molecules per call (two rounds of workgroups): pockets 25.4 / 26.7 and 23.1 / 23.4 -> binned, 500-atom molecules
    # 21.0 / 20.4 and two cfg-3 molecules (576 slabs) 19.8 / 14.5 -> one launch
    assert plan(64, 32, 2, 4000)["route"] == BINNED
    assert plan(64, 32, 2, 2000)["route"] == BINNED
    assert plan(64, 32, 2, 500)["route"] == DIRECT
    assert plan(48, 4, 2, 1000, mode="types")["route"] == DIRECT
    assert plan(64, 32, 3, 4000)["route"] == BINNED
    # ligands: 2 per call 15.7 / 10.7, 4 per call (2 048 slabs) 20.3 / 17.9 -> one launch; 8 per call 30.7 / 31.9, 16: 51 / 83,
    # 256 in one call 0.64 / 2.04 ms -> binned; four cfg-3 molecules 17.9 / 18.0 -> binned
    assert plan(64, 16, 2, 50)["route"] == DIRECT
    assert plan(64, 16, 4, 50)["route"] == DIRECT
    assert plan(64, 32, 3, 8)["route"] == DIRECT
    assert plan(64, 16, 8, 50)["route"] == BINNED
    assert plan(64, 16, 256, 50)["route"] == BINNED
    assert plan(48, 4, 4, 1000, mode="types")["route"] == BINNED
    # 96^3 and 128^3 single calls: 2 304 / 4 096 workgroups, 44 / 30 and 63 / 31 us
    assert plan(96, 32)["route"] == BINNED
    # channel-wise radii for features: always the grouped launch of the binned pipeline, chunks of 32 channels
    p = plan(64, 40, 1, 4000, radii_type="channel-wise")
    assert (p["route"], p["grouped"], p["ct"], p["ncc"], p["weights_in_place"], p["ct_rem"]) == (BINNED, 1, 32, 2, 1, 0)
    # ... but channel-wise radii for types are per-atom radii (numpy/voxelizer.py:284-285): nothing special
    assert plan(64, 8, 1, 4000, mode="types", radii_type="channel-wise")["grouped"] == 0


def test_slab_plans_for_long_rows():
    """plan_slabs, profiles/r03_odd_dimensions.txt ROW_SWEEP: rows of 65 ... 128 voxels stay whole (9 ... 16 waves) unless
    D % 32 == 0 - D = 72 1.80 -> 4.16 TB/s, 88 1.94 -> 4.02, 104 2.12 -> 4.03, 120 2.25 -> 3.81; 96 / 128 keep chunks of 8."""
    for D, nw, nzc in ((64, 8, 1), (48, 6, 1), (50, 7, 1), (65, 9, 1), (72, 9, 1), (88, 11, 1), (104, 13, 1), (120, 15, 1), (96, 8, 2),
                       (128, 8, 2), (100, 13, 1)):
        p = plan(D, 32, 16)
        assert (p["nw"], p["nzc"]) == (nw, nzc), (D, p)
    # beyond 128: rows that are not multiples of 64 B in as few equal chunks as 16 waves allow (D = 136 1.90 -> 2.10 TB/s),
    # multiples of 64 B keep chunks of 8 (D = 144 3.36 against 2.54-2.67)
    assert (plan(136, 32, 16)["nw"], plan(136, 32, 16)["nzc"]) == (9, 2)
    assert (plan(200, 16, 16)["nw"], plan(200, 16, 16)["nzc"]) == (13, 2)
    assert (plan(144, 32, 16)["nw"], plan(144, 32, 16)["nzc"]) == (8, 3)
    # the one-launch route and float64 grids keep at most 8 waves per slab
    assert plan(72, 32, 16, precision=64)["nw"] == 8


def test_channel_chunks_and_remainders():
    # C = 33: 32 on the wide kernel + 1 on the narrowest (0.66 -> 0.57 ms per 64 molecules); C = 40: 32 + 8; C = 48: 32 + 16
    for C, nfull, ct_rem, cpad in ((33, 1, 1, 36), (36, 1, 4, 36), (40, 1, 8, 40), (48, 1, 16, 48), (65, 2, 1, 68), (72, 2, 8, 72)):
        p = plan(64, C, 64, 4000)
        assert (p["ct"], p["nfull"], p["ct_rem"], p["cpad"]) == (32, nfull, ct_rem, cpad), (C, p)
    # a remainder of more than half a chunk needs the wide kernel anyway: one launch of two chunks
    p = plan(64, 50, 64, 4000)
    assert (p["ncc"], p["nfull"], p["ct_rem"], p["cpad"], p["weights_in_place"]) == (2, 2, 0, 64, 0)
    # C = 64: rows read in place
    assert plan(64, 64, 64, 4000)["weights_in_place"] == 1
    # types / single never read caller rows in place
    assert plan(64, 8, 64, 4000, mode="types")["weights_in_place"] == 0


def test_infinity_cache_chunks_and_grid_limits():
    # 256 cfg-2 molecules are one chunk (272 MB of re-read data fits the 288 MB budget); 512 / 1024 are cut (0.654 -> 0.755 of peak)
    assert plan(64, 32, 256, 4000)["nchunk"] == 1
    assert plan(64, 32, 512, 4000)["nchunk"] == 2
    assert plan(64, 32, 1024, 4000)["nchunk"] == 4
    # gridDim.y limit: 65 535 (molecule, chunk) pairs per launch
    assert plan(16, 64, 40000, 4)["nchunk"] >= 2


def test_rows_of_8k_plus_4_sub_tiles_take_chunks_of_four_in_narrow_launches():
    """D = 96 / 160 / 224: chunks of 8 sub-tiles leave a last chunk whose slabs are half empty. Launches that are bound by their
    walk (fewer than 32 channels per workgroup) take chunks of 4 (C = 16: 4.39 -> 5.91 TB/s); 32-channel chunks keep 8."""
    for C_, nw, nzc in ((4, 4, 3), (8, 4, 3), (16, 4, 3), (32, 8, 2), (64, 8, 2), (33, 8, 2)):
        p = plan(96, C_, 19, 4000)
        assert (p["nw"], p["nzc"]) == (nw, nzc), (C_, p)
    assert (plan(160, 16, 4, 4000)["nw"], plan(160, 16, 4, 4000)["nzc"]) == (4, 5)
    assert (plan(128, 16, 8, 4000)["nw"], plan(128, 16, 8, 4000)["nzc"]) == (8, 2)  # (whole chunks: nothing idle)
    assert plan(100, 16, 8, 4000)["nw"] == 13  # (rows that are not multiples of 32 voxels stay whole)


def test_long_rows_of_narrow_launches_are_cut_so_that_the_multi_sub_tile_kernel_applies():
    """One or four channels per workgroup: rows of 9 ... 15 sub-tiles in chunks of eight (nine: four; ten stay whole) on grids of
    whole 16-byte quads - C = 4 at D = 120: kernel 0.213 -> 0.110 ms; eight and more channels keep whole rows."""
    for D, nw, nzc in ((72, 4, 3), (80, 10, 1), (88, 8, 2), (100, 8, 2), (104, 8, 2), (112, 8, 2), (120, 8, 2), (128, 8, 2)):
        for C_ in (1, 3, 4):
            p = plan(D, C_, 8, 4000)
            assert (p["nw"], p["nzc"]) == (nw, nzc), (D, C_, p)
    for D, nw in ((72, 9), (88, 11), (104, 13), (120, 15)):
        assert plan(D, 8, 8, 4000)["nw"] == nw and plan(D, 16, 8, 4000)["nw"] == nw and plan(D, 32, 8, 4000)["nw"] == nw
    # forward_single on rows of five / seven sub-tiles: chunks of four (kernel 0.126 -> 0.099 ms at D = 56); four channels keep whole rows
    for D, nw1, nw4 in ((40, 4, 5), (56, 4, 7), (24, 3, 3), (48, 6, 6)):
        assert plan(D, 1, 8, 1000, mode="single")["nw"] == nw1 and plan(D, 4, 8, 1000)["nw"] == nw4, D
    assert plan(66, 4, 8, 4000)["nw"] == 9  # (rows that are not whole quads: run-wise write-out, whole rows)
    assert plan(88, 4, 8, 4000, out_aligned16=False)["nw"] == 11


def test_pacing_thresholds():
    """mvx_tuning.h: rounds paced from 49 152 workgroups (96 cfg-2 molecules: 64 molecules gain 2 % in a sustained loop and lose
    5 % in short bursts, 128 gain 3 %), empty slabs held back beyond 4 096."""
    assert plan(64, 32, 8, 4000)["pace"] == 0
    assert plan(64, 32, 16, 4000)["pace"] == 1
    assert plan(64, 32, 64, 4000)["pace"] == 1
    assert plan(64, 32, 96, 4000)["pace"] == 2
    assert plan(64, 32, 256, 4000)["pace"] == 2


def test_write_out_paths():
    # rows that are not whole 16-byte quads, or an unaligned grid slice: run-wise write-out, one slab range per XCD on whole-row grids
    p = plan(50, 32, 64)
    assert (p["vec_store"], p["xcd_ranges"]) == (0, 1)
    p = plan(64, 32, 64, out_aligned16=False)
    assert (p["vec_store"], p["xcd_ranges"]) == (0, 1)
    p = plan(130, 32, 4)
    assert (p["vec_store"], p["xcd_ranges"]) == (0, 0)
    # sub-tile edges (2, 4, 8) divide blockdim 8 / 16 / 64, not 4 / 5 / 12; a single block needs no cull at all
    for bd, lr in ((8, 0), (16, 0), (64, 0), (4, 1), (5, 1), (12, 1)):
        assert plan(64, 32, 64, blockdim=bd)["lane_range"] == lr
    assert plan(16, 8, 1, blockdim=20)["lane_range"] == 0


def test_float64_routes():
    # more than 16 channels with scalar / atom-wise radii: chunks of 32 on the matrix cores (3.65 -> 4.6 TB/s)
    p = plan(64, 32, 64, 4000, precision=64)
    assert (p["route"], p["ct"], p["ncc"]) == (F64_MX, 32, 1)
    p = plan(64, 16, 64, 4000, precision=64)
    assert (p["route"], p["ct"]) == (F64_DENSE, 16)
    p = plan(64, 32, 64, 4000, precision=64, radii_type="channel-wise")
    assert (p["route"], p["ct"], p["ncc"], p["grouped"]) == (F64_DENSE, 16, 2, 0)


def test_query_validation():
    import ctypes as C

    lib = _lib.load()
    q = _lib.MvxPlanQuery(0, 8, 32, 0, 0, 1, 1, 1, 0, 0)
    p = _lib.MvxPlan()
    assert lib.mvx_plan_call(C.byref(q), C.byref(p)) != 0
    assert lib.mvx_plan_call(None, C.byref(p)) != 0
