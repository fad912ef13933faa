"""One-off parity check on a large grid (python3 tools/big_grid_check.py [D] [N])."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import molvoxel_amd
from oracle import c_oracle

D, N = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 30000
rng = np.random.default_rng(5)
W = 0.5 * (D - 1)
xyz = rng.uniform(-W / 2 - 1, W / 2 + 1, (N, 3))
feat = rng.random((N, 3)).astype(np.float32)
rad = rng.uniform(0.8, 2.5, N).astype(np.float32)
for density in ("binary", "gaussian"):
    v = molvoxel_amd.create_voxelizer(0.5, D, "atom-wise", density, library="hip", output="numpy")
    t0 = time.time()
    out = v.forward_features(xyz, None, feat, rad)
    t1 = time.time()
    ref = c_oracle.voxelize(xyz, feat, rad, dimension=D, radii_type="atom-wise", density=density)
    print(D, N, density, "membership equal:", np.array_equal(out != 0, ref != 0), "max err", float(np.abs(out - ref).max()),
          f"gpu call {t1 - t0:.3f}s oracle {time.time() - t1:.1f}s")
