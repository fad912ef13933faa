"""Register budgets of the hot kernels, read from the built code object (no GPU needed).

The voxelize kernels live at an occupancy edge - 64 VGPRs for four 8-wave workgroups per compute unit - and a structural
edit elsewhere in the translation unit has twice pushed one variant's accumulators into scratch without any test noticing
(round 3: 193 spilled registers in the per-lane-range 32-channel variant after the slab body became an inlined function).
These bounds are what the shipped build measures plus a little slack; a change that breaks them should be looked at in
the disassembly (tools/disasm.sh) before it is accepted.
"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def res():
    from tools import regs

    if not all(os.path.exists(o) for o in regs.KERNEL_OBJECTS):
        pytest.skip("kernel objects not built (python -c 'import __graft_entry__ as g; g.build()')")
    return regs.kernel_resources()


def test_batched_voxelize_kernels_fit_their_occupancy(res):
    ks = {k: v for k, v in res.items() if k.startswith("voxelize_kernel<") and ", 512, " in k}
    assert len(ks) >= 24
    for name, r in ks.items():
        assert r["vgpr"] <= 64, (name, r)  # 8 waves per SIMD
    # the headline kernel (32 channels, gaussian, matrix-core walk) and its binary twin: no scratch in the first round
    for name in ("voxelize_kernel<32, true, false, 512, false>", "voxelize_kernel<32, false, false, 512, false>"):
        assert res[name]["scratch"] <= 32 and res[name]["vspill"] <= 6, (name, res[name])
    # narrower chunks (ligand batches, forward_types, forward_single): none at all
    for ct in (1, 4, 8, 16):
        for gauss in ("true", "false"):
            r = res[f"voxelize_kernel<{ct}, {gauss}, false, 512, false>"]
            assert r["scratch"] == 0 and r["vspill"] == 0, (ct, gauss, r)
    # per-lane-range variants (blockdim 4, 5, 12, ...) and the grouped launch (channel-wise radii): a handful of spills in
    # the cold rounds, never the accumulators
    for name, r in ks.items():
        if name.startswith("voxelize_kernel<32,"):
            assert r["vspill"] <= 20 and r["scratch"] <= 80, (name, r)  # (17: grouped + per-lane ranges, the rarest variant)


def test_whole_row_slabs_of_long_rows_keep_two_workgroups_per_unit(res):
    """Rows of 65 ... 128 voxels stay in one slab of 9 ... 16 waves (plan_slabs): the 1024-thread variants are compiled for
    64 registers so that two or three such workgroups fit a compute unit (D = 72: 4.16 against 3.75 TB/s with 128)."""
    ks = {k: v for k, v in res.items() if (k.startswith("voxelize_kernel<") and ", 1024, " in k) or k.startswith("voxelize_runs_kernel<")}
    assert len(ks) >= 24
    for name, r in ks.items():
        assert r["vgpr"] <= 64, (name, r)
    for name in ("voxelize_kernel<32, true, false, 1024, false>", "voxelize_kernel<32, false, false, 1024, false>"):
        assert res[name]["scratch"] <= 32 and res[name]["vspill"] <= 6, (name, res[name])


def test_run_wise_write_out_lives_in_its_own_kernels(res):
    """store_runs (grids whose rows are not whole 16-byte quads) is compiled into voxelize_runs_kernel and the
    per-lane-range variants only: inside the headline kernel it cost six more spilled registers and 0.5 % of its rate."""
    for maxt in (512, 1024):
        for gauss in ("true", "false"):
            r = res[f"voxelize_runs_kernel<{gauss}, {maxt}>"]
            assert r["vgpr"] <= 64 and r["vspill"] <= 12 and r["scratch"] <= 48, (gauss, maxt, r)


def test_float64_matrix_core_kernel_keeps_two_workgroups_per_unit(res):
    for name, r in res.items():
        if name.startswith("voxelize64_kernel<"):
            assert r["vgpr"] <= 128, (name, r)
            assert r["scratch"] <= 32, (name, r)  # (the Gaussian per-lane-range variant: 3 registers; the others none)
    for name in ("voxelize64_kernel<false, false, 512>", "voxelize64_kernel<false, true, 512>", "voxelize64_kernel<true, false, 512>"):
        assert res[name]["scratch"] == 0, (name, res[name])


def test_prepass_kernels_do_not_spill_vector_registers(res):
    assert res["prep_kernel"]["vspill"] == 0 and res["prep_kernel"]["scratch"] == 0
    for name, r in res.items():
        if name.startswith("xbin_kernel<"):
            assert r["vspill"] == 0 and r["scratch"] == 0, (name, r)


def test_no_per_channel_float32_kernels_are_left(res):
    """Channel-wise radii for features run grouped by radius, one table per chunk of 32 channels (at most 32 distinct radii
    per chunk): the per-channel kernels of rounds 1-3 (200 spilled registers at 32 channels) are gone, and with them every
    float32 instantiation that carried a `chanwise` parameter."""
    assert sum(k.startswith("voxelize_kernel<") for k in res) == 48  # 5 widths x {gaussian, binary} x {plain, lane ranges} x 2 sizes + 8 grouped
    # per-molecule launches: voxelize_pair_kernel alone - 5 widths x {gaussian, binary} x {no transform, transform} plus, for
    # blockdims that cut through sub-tiles (per-lane ranges), the transform-capable instantiation once more
    assert sum(k.startswith("voxelize_pair_kernel<") for k in res) == 30
    assert sum(k.startswith("voxelize_direct_kernel<") for k in res) == 0


def test_pair_kernel_fits_one_workgroup_of_sixteen_waves(res):
    """voxelize_pair_kernel runs 1024 threads (two slabs, sixteen waves) per workgroup: 128 registers per lane at most, and
    no scratch at all in the variants without a transform (the cold multi-round / multi-segment code uses lighter scan and
    stage variants beside the live accumulators), a few dozen bytes with one. The disassembly
    (tools/disasm.sh voxelize_pair_kernelILi32ELb1ELb0E molvoxel_amd/csrc/mvx_pair.o) shows where the scratch accesses sit."""
    ks = {k: v for k, v in res.items() if k.startswith("voxelize_pair_kernel<")}
    for name, r in ks.items():
        assert r["vgpr"] <= 128, (name, r)
        no_transform = ", false, false>" in name  # <CT, gauss, transform, lane ranges>
        assert r["scratch"] <= (0 if no_transform else 128), (name, r)  # (452 B once cost the cfg-2 call 30 %: profiles/r04_single_calls.txt)


def test_narrow_kernels_hold_their_accumulator_sets_in_registers(res):
    """voxelize_narrow_kernel (1 ... 8 channels, two or four sub-tiles per wave): the NSUB accumulator sets and the eight row
    loads in flight stay in registers, and since the staging became branch-free with one add per slot every variant fits the
    64 registers of 8 waves per SIMD (it needed 66 ... 84 and was compiled for 7 / 6 / 5 before; profiles/r04_narrow.txt)."""
    ks = {k: v for k, v in res.items() if k.startswith("voxelize_narrow_kernel<")}
    assert len(ks) == 10  # {1, 4} channels x {2, 4} sub-tiles + 8 channels x 2, Gaussian and binary
    for name, r in ks.items():
        assert r["vspill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 64, (name, r)


