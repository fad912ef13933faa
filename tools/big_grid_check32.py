"""One-off parity check of the 32-channel (matrix-core) path on large grids: python3 tools/big_grid_check32.py [D] [N] [precision]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import molvoxel_amd
from oracle import c_oracle, numpy_port

D, N = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 12000
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(7)
W = 0.5 * (D - 1)
xyz = rng.uniform(-W / 2 - 1, W / 2 + 1, (N, 3))
feat = (rng.random((N, 32)) - 0.3).astype(np.float32 if prec == 32 else np.float64)
rad = rng.uniform(0.8, 2.2, N).astype(np.float32 if prec == 32 else np.float64)
for density in ("binary", "gaussian"):
    v = molvoxel_amd.create_voxelizer(0.5, D, "atom-wise", density, library="hip", output="numpy", sigma=0.8, precision=prec)
    out = v.forward_features(xyz, None, feat, rad)
    if prec == 32:
        ref = c_oracle.voxelize(xyz, feat, rad, dimension=D, radii_type="atom-wise", density=density, sigma=0.8)
    else:
        ref = numpy_port.voxelize(numpy_port.GridSpec(0.5, D), xyz, feat, rad, radii_type="atom-wise", density=density, sigma=0.8, precision=64)
    print(D, N, prec, density, "membership equal:", bool(np.array_equal(out != 0, ref != 0)), "max err", float(np.abs(out - ref).max()),
          "max |ref|", float(np.abs(ref).max()))
