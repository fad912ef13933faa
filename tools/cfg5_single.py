"""One cfg-5 molecule (N = 10 000, 128^3, C = 32, per-atom radii) per forward() call: us per call, binned and direct route.

    python3 tools/cfg5_single.py [lib.so] [direct modes, default "0"] [nw] [batch]
"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molvoxel_amd.voxelizer.hip import _lib as _l
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _l.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import molvoxel_amd
from molvoxel_amd import workloads as W
wl = W.cfg5()
modes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if B > 1:
    wl = W.cfg5(batch=B)
for direct in modes:
    vox = molvoxel_amd.create_voxelizer(0.5, 128, "atom-wise", "gaussian", library="hip", sigma=1.0)
    vox.debug_option("direct", direct)
    if NW: vox.debug_option("nw", NW)
    if os.environ.get("CFG5_OVERLAP"): vox.set_overlap_prepass(True)  # opt-in: pre-pass of call k+1 under call k's voxelize launch
    c = vox.asarray(np.concatenate(wl.coords), "coords"); f = vox.asarray(np.concatenate(wl.channels), "features")
    r = vox.asarray(np.concatenate(wl.radii), "radii")
    off = np.arange(B + 1, dtype=np.int64) * wl.coords[0].shape[0]
    g = vox.get_empty_grid(32, batch_size=B) if B > 1 else vox.get_empty_grid(32)
    call = (lambda: vox.forward_batch(c, off, None, f, r, out_grid=g)) if B > 1 else (lambda: vox.forward(c, None, f, r, out_grid=g))
    for _ in range(50): call()
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200 // B): call()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / (200 // B))
    el = float(np.median(best))
    ab = wl.algorithmic_bytes(0) * B
    print(f"{os.path.basename(_l.LIB_PATH)} direct {direct} nw {NW} batch {B}: {el*1e6:.1f} us/call (min {min(best)*1e6:.1f}), {ab/el/1e12:.2f} TB/s = {ab/el/8e12:.3f} of peak")
