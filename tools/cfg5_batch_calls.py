"""Back-to-back forward_batch calls over B cfg-5 molecules, for rocprofv3 --kernel-trace + tools/call_gaps.py.

    python3 tools/cfg5_batch_calls.py B [calls]
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
from molvoxel_amd import workloads as W
B = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = W.cfg5(batch=B)
vox = molvoxel_amd.create_voxelizer(0.5, 128, "atom-wise", "gaussian", library="hip", sigma=1.0)
c = vox.asarray(np.concatenate(wl.coords), "coords"); f = vox.asarray(np.concatenate(wl.channels), "features")
r = vox.asarray(np.concatenate(wl.radii), "radii")
off = np.arange(B + 1, dtype=np.int64) * wl.coords[0].shape[0]
g = vox.get_empty_grid(32, batch_size=B)
for _ in range(calls):
    vox.forward_batch(c, off, None, f, r, out_grid=g)
torch.cuda.synchronize()
