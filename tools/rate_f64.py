"""float64 grids on cfg-2 geometry:  python3 tools/rate_f64.py [molecules] [channels per workgroup, e.g. 16,32]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
if len(sys.argv) > 3:  # another build of the library (A/B)
    import os
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.abspath(sys.argv[3])
import molvoxel_amd

B, N, D, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 4000, 64, int(os.environ.get("F64_C", "32"))
rng = np.random.default_rng(0)
W = 0.5 * (D - 1)
coords = rng.uniform(-W / 2, W / 2, (B * N, 3))
offsets = np.arange(B + 1, dtype=np.int64) * N
CT64 = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32").split(",")]  # channels per workgroup: 32 (one chunk) | 16
for density, ct64 in [(d, c) for d in ("gaussian", "binary") for c in CT64]:
    vox = molvoxel_amd.create_voxelizer(0.5, D, "scalar", density, library="hip", precision=64)
    vox.debug_option("max_ct64", ct64)
    if os.environ.get("F64_NW"):
        vox.debug_option("nw", int(os.environ["F64_NW"]))
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
    vox.set_profiling(False)
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    torch.cuda.synchronize()
    call = (time.perf_counter() - t0) / 10 * 1e3
    print(f"f64 {density} ct {ct64}: kernel {ms:.3f} ms for {B} molecules = {ms / B * 1e3:.0f} us/molecule, {B * 8 * C * D**3 / ms / 1e6:.0f} GB/s; "
          f"whole call {call:.3f} ms = {B * 8 * C * D**3 / call / 1e6:.0f} GB/s")
