"""Whole-call rates of operator variants outside the BASELINE configurations (cfg-2 geometry unless stated, 64 molecules per
call): looks for performance cliffs.   python3 tools/variants.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):  # another build of the library (A/B): path relative to the repo root
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])
import molvoxel_amd
ONLY = os.environ.get("ONLY")  # substring filter over the row names

B, N = 64, 4000
rng = np.random.default_rng(0)


def run(name, D, radii_type, density, mode, C, radii, res=0.5, transform=False, N=N, **kw):
    if ONLY and ONLY not in name:
        return
    W = res * (D - 1)
    vox = molvoxel_amd.create_voxelizer(res, D, radii_type, density, library="hip", **kw)
    xyz = rng.uniform(-W / 2, W / 2, (B * N, 3))
    coords = vox.asarray(xyz, "coords")
    if mode == "features":
        chan = vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
    elif mode == "types":
        t = rng.integers(0, C, B * N)
        t[::N] = C - 1
        chan = vox.asarray(t, "types")
    else:
        chan = None
    if isinstance(radii, np.ndarray):
        radii = vox.asarray(radii, "radii")
    off = np.arange(B + 1, dtype=np.int64) * N
    Cout = 1 if mode == "single" else C
    out = vox.get_empty_grid(Cout, batch_size=B)
    centers = np.zeros((B, 3)) if transform else None
    kwargs = dict(random_translation=1.0, random_rotation=True) if transform else {}
    call = lambda: vox.forward_batch(coords, off, centers, chan, radii, out_grid=out, **kwargs)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20
    nbytes = B * Cout * D**3 * (8 if kw.get("precision") == 64 else 4)
    print(f"{name:58s} {el*1e3:8.3f} ms/call  {nbytes/el/1e12:5.2f} TB/s of grid bytes  {B/el:9.0f} molecules/s")


r_atom = rng.uniform(0.8, 1.2, B * N).astype(np.float32)
run("cfg-2 (features C=32, scalar r=1.0, gaussian)", 64, "scalar", "gaussian", "features", 32, 1.0)
run("binary instead of gaussian", 64, "scalar", "binary", "features", 32, 1.0)
run("atom-wise radii 0.8-1.2", 64, "atom-wise", "gaussian", "features", 32, r_atom)
run("channel-wise radii 0.8-1.2 (features)", 64, "channel-wise", "gaussian", "features", 32, rng.uniform(0.8, 1.2, 32).astype(np.float32))
run("channel-wise radii, 32 channels, 4 distinct radii", 64, "channel-wise", "gaussian", "features", 32, np.repeat(np.float32([0.85, 0.95, 1.05, 1.15]), 8))
run("channel-wise radii, 32 channels, 1 radius", 64, "channel-wise", "gaussian", "features", 32, np.full(32, 1.0, np.float32))
run("forward_types, 8 types, gaussian", 64, "scalar", "gaussian", "types", 8, 1.0)
run("forward_types, 32 types, binary", 64, "scalar", "binary", "types", 32, 1.0)
run("forward_types, 8 types, channel-wise radii", 64, "channel-wise", "gaussian", "types", 8, rng.uniform(0.8, 1.2, 8).astype(np.float32))
run("forward_single", 64, "scalar", "gaussian", "single", 1, 1.0)
run("features C=5", 64, "scalar", "gaussian", "features", 5, 1.0)
run("features C=64", 64, "scalar", "gaussian", "features", 64, 1.0)
run("features C=33", 64, "scalar", "gaussian", "features", 33, 1.0)
run("D=50 (not a multiple of 4: run-wise write-out)", 50, "scalar", "gaussian", "features", 32, 1.0)
run("D=48", 48, "scalar", "gaussian", "features", 32, 1.0)
run("D=48 at cfg-2 density (1 688 atoms per molecule)", 48, "scalar", "gaussian", "features", 32, 1.0, N=1688)
run("D=96", 96, "scalar", "gaussian", "features", 16, 1.0)
run("blockdim=5 (sub-tiles straddle reference blocks)", 64, "scalar", "gaussian", "features", 32, 1.0, blockdim=5)
run("blockdim=64 (no block cull)", 64, "scalar", "gaussian", "features", 32, 1.0, blockdim=64)
run("random rotation + translation per molecule", 64, "scalar", "gaussian", "features", 32, 1.0, transform=True)
run("resolution 1.0, radius 2.0", 64, "scalar", "gaussian", "features", 32, 2.0, res=1.0)
run("precision=64", 64, "scalar", "gaussian", "features", 32, 1.0, precision=64)
