"""Voxelize-kernel rate on cfg-2 geometry for several channel counts (python3 tools/rate_vs_channels.py)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import molvoxel_amd

B, N, D = 64, 4000, 64
rng = np.random.default_rng(0)
W = 0.5 * (D - 1)
coords = rng.uniform(-W / 2, W / 2, (B * N, 3))
offsets = np.arange(B + 1, dtype=np.int64) * N
for C in (8, 16, 32, 64):
    vox = molvoxel_amd.create_voxelizer(0.5, D, library="hip")
    dc = vox.asarray(coords, "coords")
    df = vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
    out = vox.get_empty_grid(C, batch_size=B)
    for _ in range(3):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    vox.set_profiling(True)
    for _ in range(10):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    torch.cuda.synchronize()
    t = vox.read_kernel_times_ms()
    ms = float(np.sum(t)) / 10
    alg = B * (4 * C * D**3 + N * (24 + 4 * C + 4))
    print(f"C={C:3d} kernel {ms:.3f} ms  {alg / ms / 1e6:.0f} GB/s  launches/step {len(t) // 10}")
    del out
