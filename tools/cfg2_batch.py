"""cfg-2 batches through forward_batch with debug options (kernel time by HIP events and whole-call time):

    python3 tools/cfg2_batch.py [batch] [option=value ...]      e.g. python3 tools/cfg2_batch.py 256 nw=4 radius=1.5
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):  # another build of the library (A/B): path relative to the repo root
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])
import molvoxel_amd
from molvoxel_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = W.cfg2(batch=B)
vox = molvoxel_amd.create_voxelizer(0.5, 64, library="hip")
radius = 1.0
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "radius":  # scalar radius in Angstrom instead of cfg-2's 1.0 (larger radii: more candidates per slab)
        radius = float(v)
    else:
        vox.debug_option(k, int(v))
coords = vox.asarray(np.concatenate(wl.coords), "coords")
feats = vox.asarray(np.concatenate(wl.channels), "features")
offsets = np.arange(B + 1, dtype=np.int64) * 4000
out = vox.get_empty_grid(32, batch_size=B)
for _ in range(25):
    vox.forward_batch(coords, offsets, None, feats, radius, out_grid=out)
torch.cuda.synchronize()
vox.set_profiling(True)
t0 = time.perf_counter()
for _ in range(40):
    vox.forward_batch(coords, offsets, None, feats, radius, out_grid=out)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 40
k = np.array(vox.read_kernel_times_ms())
ab = wl.algorithmic_bytes(0) * B
sumtxt = f"  checksum {int(out.view(torch.int32).to(torch.int64).sum().item())}" if os.environ.get("SUM") else ""  # (A/B builds must agree bit for bit)
print(f"cfg-2 x {B} {' '.join(sys.argv[2:])}: kernel {np.median(k)*1e3:.1f} us ({ab/np.median(k)/1e9/8:.3f} of peak), step {el*1e3:.4f} ms ({ab/el/8e12:.3f}){sumtxt}")
