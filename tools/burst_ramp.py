"""From an idle GPU: ms per call in consecutive groups of 5 calls (how long until the sustained rate is reached?)
    LIB=... python3 tools/burst_ramp.py [batch] [groups] [idle seconds]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])
import molvoxel_amd
from molvoxel_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
wl = W.cfg2(batch=B)
vox = molvoxel_amd.create_voxelizer(0.5, 64, library="hip")
coords = vox.asarray(np.concatenate(wl.coords), "coords")
feats = vox.asarray(np.concatenate(wl.channels), "features")
offsets = np.arange(B + 1, dtype=np.int64) * 4000
out = vox.get_empty_grid(32, batch_size=B)
for _ in range(3):
    vox.forward_batch(coords, offsets, None, feats, 1.0, out_grid=out)
torch.cuda.synchronize()
acc = np.zeros(G)
R = 4
for rep in range(R):
    time.sleep(idle)
    for g in range(G):
        t0 = time.perf_counter()
        for _ in range(5):
            vox.forward_batch(coords, offsets, None, feats, 1.0, out_grid=out)
        torch.cuda.synchronize()
        acc[g] += (time.perf_counter() - t0) / 5 * 1e3
print(f"cfg-2 x {B} after {idle} s idle, groups of 5 calls, ms per call: " + " ".join(f"{x / R:.3f}" for x in acc))
