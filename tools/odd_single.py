"""Per-molecule calls on grids whose rows are not whole 16-byte quads (odd dimensions, unaligned slices of a batch grid) and
on blockdims that cut through sub-tiles: us per call, one launch (default route) against the binned pipeline.
    python3 tools/odd_single.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import molvoxel_amd

rng = np.random.default_rng(0)


def timed(call, n=200):
    for _ in range(20): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for D, C, blockdim, unaligned in ((49, 32, None, False), (33, 32, None, False), (63, 32, None, False), (50, 16, None, False), (48, 32, None, True),
                                  (64, 32, None, True), (64, 32, 5, False), (48, 8, 12, False)):
    N = int(round(4000 * ((D - 1) / 63.0) ** 3))
    W = 0.5 * (D - 1)
    xyz = rng.uniform(-W / 2, W / 2, (N, 3))
    res = []
    for route in (0, -1):
        kw = {"blockdim": blockdim} if blockdim else {}
        v = molvoxel_amd.create_voxelizer(0.5, D, library="hip", **kw)
        v.debug_option("direct", route)
        c, f = v.asarray(xyz, "coords"), v.asarray(rng.random((N, C)).astype(np.float32), "features")
        if unaligned:  # slice 1 of a batch grid whose slices start 4 bytes off a 16-byte boundary
            flat = torch.empty(2 * C * D ** 3 + 1, dtype=torch.float32, device=v.device)
            g = flat[1:1 + C * D ** 3].view(C, D, D, D)
        else:
            g = v.get_empty_grid(C)
        res.append(timed(lambda: v.forward(c, None, f, 1.0, out_grid=g)))
    print(f"D={D:3d} C={C:3d} N={N:5d} blockdim={blockdim} {'unaligned grid' if unaligned else ''}: binned {res[0]:6.1f} us, default route {res[1]:6.1f} us", flush=True)
