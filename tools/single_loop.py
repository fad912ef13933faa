"""forward_single (or C narrow channels) in a loop, for rocprofv3 / tools/pmc_cmd.sh:  python3 tools/single_loop.py [C] [density]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
density = sys.argv[2] if len(sys.argv) > 2 else "gaussian"
B, N, D = 64, 4000, 64
rng = np.random.default_rng(0)
W = 0.5 * (D - 1)
vox = molvoxel_amd.create_voxelizer(0.5, D, "scalar", density, library="hip")
vox.debug_option("direct", 0)
coords = vox.asarray(rng.uniform(-W / 2, W / 2, (B * N, 3)), "coords")
chan = None if C == 1 else vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
off = np.arange(B + 1, dtype=np.int64) * N
out = vox.get_empty_grid(C, batch_size=B)
for _ in range(40):
    vox.forward_batch(coords, off, None, chan, 1.0, out_grid=out)
torch.cuda.synchronize()
