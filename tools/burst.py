"""Short bursts of batched calls from a cold start (a few calls, then idle): does a setting tuned in a sustained loop hold?
    LIB=... python3 tools/burst.py [batch] [warm-up calls] [timed calls]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])
import molvoxel_amd
from molvoxel_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wl = W.cfg2(batch=B)
vox = molvoxel_amd.create_voxelizer(0.5, 64, library="hip")
coords = vox.asarray(np.concatenate(wl.coords), "coords")
feats = vox.asarray(np.concatenate(wl.channels), "features")
offsets = np.arange(B + 1, dtype=np.int64) * 4000
out = vox.get_empty_grid(32, batch_size=B)
res = []
for rep in range(6):
    time.sleep(0.5)  # idle: clocks fall back
    for _ in range(warm):
        vox.forward_batch(coords, offsets, None, feats, 1.0, out_grid=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        vox.forward_batch(coords, offsets, None, feats, 1.0, out_grid=out)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / n * 1e3)
print(f"cfg-2 x {B}, bursts of {warm} + {n} calls after 0.5 s idle: ms per call " + " ".join(f"{x:.4f}" for x in res) + f"  median {np.median(res):.4f}")
