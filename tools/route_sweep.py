"""One-launch route (voxelize_pair_kernel) against the binned pipeline, us per call, over molecule size, grid, channels and
molecules per call - the data behind plan_call's route limits (mvx_tuning.h).   python3 tools/route_sweep.py [single|multi|all]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import molvoxel_amd

rng = np.random.default_rng(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(v, call, n=150):
    for _ in range(20): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def one(D, N, C, B=1, ligand=False):
    W = 0.5 * (D - 1)
    res = []
    for route in (0, 1):
        v = molvoxel_amd.create_voxelizer(0.5, D, library="hip")
        v.debug_option("direct", route)
        if B == 1:
            xyz = rng.normal(0, 2.0, (N, 3)) if ligand else rng.uniform(-W / 2, W / 2, (N, 3))
            c, f = v.asarray(xyz, "coords"), v.asarray(rng.random((N, C)).astype(np.float32), "features")
            g = v.get_empty_grid(C)
            res.append(timed(v, lambda: v.forward(c, None, f, 1.0, out_grid=g)))
        else:
            xyz = [rng.normal(0, 2.0, (N, 3)) if ligand else rng.uniform(-W / 2, W / 2, (N, 3)) for _ in range(B)]
            feats = [rng.random((N, C)).astype(np.float32) for _ in range(B)]
            co = v.asarray(np.concatenate(xyz), "coords")
            fe = v.asarray(np.concatenate(feats), "features")
            offsets = np.arange(B + 1, dtype=np.int64) * N
            g = v.get_empty_grid(C, batch_size=B)
            res.append(timed(v, lambda: v.forward_batch(co, offsets, None, fe, 1.0, out_grid=g)))
    ncc = (C + 31) // 32
    slabs = ((D + 1) // 2) * ((D + 3) // 4) * ncc * B
    print(f"D={D:3d} N={N:6d} C={C:3d} B={B:3d}{' ligand' if ligand else '       '}: slabs {slabs:6d}, atom tests {slabs * N / 1e6:7.2f} M: "
          f"binned {res[0]:7.1f} us, one launch {res[1]:7.1f} us  -> {'ONE' if res[1] < res[0] else 'binned'}", flush=True)


if what in ("single", "all"):
    for N in (50, 500, 2000, 4000, 8000, 12000, 16000, 24000, 32000, 48000):
        one(64, N, 32)
    for D in (32, 48, 56):
        for N in (500, 4000, 16000):
            one(D, N, 32)
    for C in (4, 16):
        for N in (500, 2000, 4000, 8000):
            one(64, N, C)
if what in ("chunks", "all"):  # channel chunks (C > 32): a loop inside the pair kernel's workgroups
    for C in (33, 40, 64, 96, 128):
        for N in (50, 500, 2000, 4000, 8000):
            one(64, N, C)
    one(48, 1700, 64)
    one(32, 500, 64)
    for B in (2, 3):
        for N in (50, 500, 4000):
            one(64, N, 64, B)
if what in ("multi", "all"):
    for B in (2, 3, 4, 6, 8):
        for N in (500, 2000, 4000):
            one(64, N, 32, B)
    for B in (2, 4, 8, 16, 32):
        one(64, 50, 16, B, ligand=True)
    for B in (2, 4, 8):
        one(48, 1000, 4, B)
