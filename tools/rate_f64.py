"""float64 grids on cfg-2 geometry (python3 tools/rate_f64.py)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import molvoxel_amd

B, N, D, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 4000, 64, 32
rng = np.random.default_rng(0)
W = 0.5 * (D - 1)
coords = rng.uniform(-W / 2, W / 2, (B * N, 3))
offsets = np.arange(B + 1, dtype=np.int64) * N
for density in ("gaussian", "binary"):
    vox = molvoxel_amd.create_voxelizer(0.5, D, "scalar", density, library="hip", precision=64)
    dc = vox.asarray(coords, "coords")
    df = vox.asarray(rng.random((B * N, C)), "features")
    out = vox.get_empty_grid(C, batch_size=B)
    for _ in range(2):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    vox.set_profiling(True)
    for _ in range(5):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    torch.cuda.synchronize()
    ms = float(np.sum(vox.read_kernel_times_ms())) / 5
    print(f"f64 {density}: kernel {ms:.3f} ms for {B} molecules = {ms / B * 1e3:.0f} us/molecule, {B * 8 * C * D**3 / ms / 1e6:.0f} GB/s")
