"""Where a single forward() call spends its host time (python3 tools/latency.py)."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import molvoxel_amd

pc = np.load("tests/golden/pointcloud_10gs.npz")
vox = molvoxel_amd.create_voxelizer(0.5, 64, library="hip")
coords = vox.asarray(pc["ligand_xyz"] - pc["ligand_xyz"].mean(0), "coords")
feats = vox.asarray(pc["ligand_feat5"], "features")
grid = vox.get_empty_grid(5)
for _ in range(100):
    vox.forward(coords, None, feats, 1.0, out_grid=grid)
torch.cuda.synchronize()
n = 5000
t0 = time.perf_counter()
for _ in range(n):
    vox.forward(coords, None, feats, 1.0, out_grid=grid)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per call {1e6 * (t1 - t0) / n:.1f} us, incl. drain {1e6 * (t2 - t0) / n:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    vox.forward(coords, None, feats, 1.0, out_grid=grid)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
