"""Narrow chunks (forward_single, few types, C <= 16, the remainder of C = 33): time per call and a checksum of the grids
(A/B builds must agree bit for bit).   LIB=molvoxel_amd/csrc/ab/libmvx_x.so python3 tools/narrow_probe.py
(cfg-2 geometry, 64 molecules per call)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["LIB"])
import molvoxel_amd

B, N, D = 64, 4000, 64
rng = np.random.default_rng(0)
W = 0.5 * (D - 1)
xyz = rng.uniform(-W / 2, W / 2, (B * N, 3))
off = np.arange(B + 1, dtype=np.int64) * N
for mode, C, density, radius in (("single", 1, "gaussian", 1.0), ("single", 1, "binary", 1.0), ("single", 1, "gaussian", 1.5),
                                 ("types", 4, "gaussian", 1.0), ("types", 8, "binary", 1.0), ("features", 5, "gaussian", 1.0),
                                 ("features", 8, "gaussian", 1.5), ("features", 16, "gaussian", 1.0), ("features", 33, "gaussian", 1.0),
                                 ("features", 40, "gaussian", 1.0), ("features", 32, "gaussian", 1.0)):
    vox = molvoxel_amd.create_voxelizer(0.5, D, "scalar", density, library="hip")
    vox.debug_option("direct", 0)
    coords = vox.asarray(xyz, "coords")
    r2 = np.random.default_rng(C)
    chan = None
    if mode == "features":
        chan = vox.asarray(r2.random((B * N, C)).astype(np.float32), "features")
    elif mode == "types":
        t = r2.integers(0, C, B * N); t[::N] = C - 1
        chan = vox.asarray(t, "types")
    out = vox.get_empty_grid(C, batch_size=B)
    out.fill_(float("nan"))
    call = lambda: vox.forward_batch(coords, off, None, chan, radius, out_grid=out)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{mode:8s} C = {C:2d} {density:8s} r = {radius}: {ms:.3f} ms per {B} molecules  checksum {int(out.view(torch.int32).to(torch.int64).sum().item())}")
