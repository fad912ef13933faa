"""From an idle GPU: the voxelize kernel's own time (HIP events) call by call - is the slow start the kernel or the host?
    LIB=... python3 tools/burst_kernel.py [batch] [calls] [idle seconds]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])
import molvoxel_amd
from molvoxel_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
vox.set_profiling(True)
for rep in range(3):
    time.sleep(idle)
    t0 = time.perf_counter()
    for _ in range(n):
        vox.forward_batch(coords, offsets, None, feats, 1.0, out_grid=out)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n * 1e3
    k = np.array(vox.read_kernel_times_ms())
    print(f"after {idle} s idle, {n} calls, {el:.3f} ms per call; kernel ms call by call: " + " ".join(f"{x:.3f}" for x in k))
