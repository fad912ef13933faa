"""Narrow channel chunks in batch form: whole-call time, voxelize-kernel time (HIP events, summed over a call's voxelize
launches) and TB/s of grid bytes, plus a checksum of the grids (A/B builds must agree bit for bit).

    python3 tools/narrow_rows.py [row ...]        rows: single types8 feat5 feat16 cfg3x256 cfg1x256 (default: all)
    LIB=molvoxel_amd/csrc/ab/libmvx_x.so python3 tools/narrow_rows.py single      another build of the library

single / types8 / feat5 / feat16: cfg-2 geometry (64^3, 4000 atoms per molecule, r = 1.0, Gaussian), 64 molecules per call;
cfg3x256: BASELINE cfg-3 (binary forward_types, 4 channels, 48^3, N = 1000) x 256; cfg1x256: cfg-1's channel count (C = 5,
64^3) at pocket density x 256."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["LIB"])
import molvoxel_amd
from molvoxel_amd import workloads as W

ROWS = {  # name: (mode, C, density, D, N, B)
    "single": ("single", 1, "gaussian", 64, 4000, 64),
    "types8": ("types", 8, "gaussian", 64, 4000, 64),
    "feat5": ("features", 5, "gaussian", 64, 4000, 64),
    "feat16": ("features", 16, "gaussian", 64, 4000, 64),
    "cfg3x256": ("types", 4, "binary", 48, 1000, 256),
    "cfg1x256": ("features", 5, "gaussian", 64, 4000, 256),
    "lig8x128": ("types", 8, "gaussian", 64, 50, 128),  # ligand-sized molecules: mostly empty slabs
}
CALLS = int(os.environ.get("CALLS", 30))


def run(name):
    mode, C, density, D, N, B = ROWS[name]
    vox = molvoxel_amd.create_voxelizer(0.5, D, "scalar", density, library="hip")
    if os.environ.get("NARROW_SUB"):  # A/B: sub-tiles per wave of narrow chunks (1: voxelize_kernel; 2 | 4: voxelize_narrow_kernel)
        vox.debug_option("narrow_sub", int(os.environ["NARROW_SUB"]))
    if name == "cfg3x256":
        wl = W.cfg3(batch=B)
        xyz = np.concatenate(wl.coords)
        chan = vox.asarray(np.concatenate(wl.channels), "types")
    else:
        rng = np.random.default_rng(0)
        Wd = 0.5 * (D - 1)
        xyz = rng.uniform(-Wd / 2, Wd / 2, (B * N, 3))
        r2 = np.random.default_rng(C)
        chan = None
        if mode == "features":
            chan = vox.asarray(r2.random((B * N, C)).astype(np.float32), "features")
        elif mode == "types":
            t = r2.integers(0, C, B * N); t[::N] = C - 1
            chan = vox.asarray(t, "types")
    coords = vox.asarray(xyz, "coords")
    off = np.arange(B + 1, dtype=np.int64) * N
    out = vox.get_empty_grid(C, batch_size=B)
    out.fill_(float("nan"))
    call = lambda: vox.forward_batch(coords, off, None, chan, 1.0, num_channels=C if mode == "types" else None, out_grid=out)
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(CALLS):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / CALLS * 1e3
    vox.set_profiling(True)
    for _ in range(CALLS):
        call()
    torch.cuda.synchronize()
    k_ms = float(np.sum(vox.read_kernel_times_ms())) / CALLS
    vox.set_profiling(False)
    gb = B * C * D**3 * 4
    print(f"{name:9s} {mode:8s} C = {C:2d} {density:8s} D = {D} x {B}: call {ms:.3f} ms ({gb / ms / 1e9:.2f} TB/s of grid bytes), "
          f"voxelize kernel {k_ms:.3f} ms ({gb / k_ms / 1e9:.2f} TB/s)  checksum {int(out.view(torch.int32).to(torch.int64).sum().item())}", flush=True)


for n in (sys.argv[1:] or list(ROWS)):
    run(n)
