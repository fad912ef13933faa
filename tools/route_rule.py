"""Direct vs binned for single molecules outside the default rule (python3 tools/route_rule.py)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import molvoxel_amd
rng = np.random.default_rng(0)
for D, N, C in ((128, 40, 5), (128, 1000, 8), (128, 3000, 16), (96, 40, 5), (96, 2000, 16), (64, 40, 64), (64, 2000, 64), (64, 12000, 32)):
    W = 0.5 * (D - 1)
    xyz = rng.normal(0, 2.0, (N, 3)) if N < 100 else rng.uniform(-W / 2, W / 2, (N, 3))
    res = []
    for route in (0, 1):
        v = molvoxel_amd.create_voxelizer(0.5, D, library="hip")
        v.debug_option("direct", route)
        c, f = v.asarray(xyz, "coords"), v.asarray(rng.random((N, C)).astype(np.float32), "features")
        g = v.get_empty_grid(C)
        for _ in range(30): v.forward(c, None, f, 1.0, out_grid=g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): v.forward(c, None, f, 1.0, out_grid=g)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    ncc = (C + 31) // 32
    wgs = ((D + 1) // 2) * ((D + 3) // 4) * ((D + 63) // 64) * ncc
    print(f"D={D} N={N} C={C}: workgroups {wgs}, atom tests {wgs * N / 1e6:.1f} M: binned {res[0]:.1f} us, direct {res[1]:.1f} us")
