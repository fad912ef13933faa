"""Small channel counts at cfg-2 geometry (64 molecules per call) with the plan's slabs or forced waves per slab:
python3 tools/small_c_probe.py   (debug option "nw")"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
B, N = 64, 4000
rng = np.random.default_rng(0)
W = 0.5 * 63
xyz = rng.uniform(-W / 2, W / 2, (B * N, 3))
off = np.arange(B + 1, dtype=np.int64) * N
for C, mode in ((1, "single"), (4, "types"), (8, "types"), (8, "features"), (16, "features"), (32, "features")):
    for nw in (0, 4, 2):
        vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip")
        if nw: vox.debug_option("nw", nw)
        coords = vox.asarray(xyz, "coords")
        chan = None if mode == "single" else (vox.asarray(rng.integers(0, C, B * N), "types") if mode == "types" else vox.asarray(rng.random((B * N, C)).astype(np.float32), "features"))
        out = vox.get_empty_grid(C, batch_size=B)
        call = lambda: vox.forward_batch(coords, off, None, chan, 1.0, num_channels=C if mode == "types" else None, out_grid=out)
        for _ in range(5): call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): call()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
        print(f"C={C:2d} {mode:8s} nw={nw}: {el*1e3:.3f} ms/call")
