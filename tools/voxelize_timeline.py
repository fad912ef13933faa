"""Per-phase timeline of the batched voxelize_kernel from a -DMVX_DIAG build (tools/ab_build.sh diag "-DMVX_DIAG").

    python3 tools/voxelize_timeline.py [batch | cfg5] [lib] [cfg-5 batch]     (cfg-2 molecules, default 256; or cfg-5 molecules)
Stamps per workgroup (s_memtime = shader cycles; only deltas inside a workgroup are meaningful):
  0 start | 1 line arrived | 2 rows staged by wave 0 | 3 staging barrier passed | 8+w walk end of wave w |
  4 barrier after the walk | 5 write-out round 0 (4 channels) done | 6 all stores issued | 7 = candidates in the line
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molvoxel_amd.voxelizer.hip import _lib as _l

_l.LIB_PATH = os.path.join(ROOT, sys.argv[2] if len(sys.argv) > 2 else "molvoxel_amd/csrc/ab/libmvx_diag.so")
_l.SIGNATURES["mvx_debug_read_diag"] = (C.c_int, [_l.Handle, C.c_void_p, C.c_int64])
import molvoxel_amd
from molvoxel_amd import workloads as W

CFG5 = len(sys.argv) > 1 and sys.argv[1] == "cfg5"  # one cfg-5 molecule (N = 10 000, 128^3), binned route
B = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) if CFG5 else (int(sys.argv[1]) if len(sys.argv) > 1 else 256)
if CFG5:
    wl = W.cfg5(batch=B)
    vox = molvoxel_amd.create_voxelizer(0.5, 128, "atom-wise", "gaussian", library="hip", sigma=1.0)
    vox.debug_option("direct", 0)
    radii = vox.asarray(np.concatenate(wl.radii), "radii")
    nwg = 4096 * B
else:
    wl = W.cfg2(batch=B)
    vox = molvoxel_amd.create_voxelizer(0.5, 64, library="hip")
    radii = float(os.environ.get("RADIUS", "1.0"))  # scalar radius in Angstrom (cfg-2 itself: 1.0)
    nwg = B * 512
    vox.debug_option("direct", 0)
coords = vox.asarray(np.concatenate(wl.coords[:B]), "coords")
NCH = int(os.environ.get("CHANNELS", "32"))  # cfg-2 with fewer feature channels (1: forward_single, no features at all)
feats = None if NCH == 1 else vox.asarray(np.concatenate(wl.channels[:B])[:, :NCH].copy(), "features")
offsets = np.arange(B + 1, dtype=np.int64) * wl.coords[0].shape[0]
out = vox.get_empty_grid(NCH if not CFG5 else 32, batch_size=B)
for _ in range(25):
    vox.forward_batch(coords, offsets, None, feats, radii, out_grid=out)
torch.cuda.synchronize()
vox.debug_option("vk_stamps", nwg)
vox.forward_batch(coords, offsets, None, feats, radii, out_grid=out)
buf = np.zeros((nwg, 16), dtype=np.uint64)
_l.check(vox._lib.mvx_debug_read_diag(vox._handle, buf.ctypes.data, buf.nbytes))
vox.debug_option("vk_stamps", 0)
t = buf.astype(np.float64)
n = t[:, 7]
ok = (t[:, 6] > 0) & (n > 0)
t = t[ok]
kc = lambda a, b: (t[:, b] - t[:, a]) / 1000.0
def line(name, x):
    print(f"{name:34s} p10 {np.percentile(x, 10):6.2f}  p50 {np.percentile(x, 50):6.2f}  p90 {np.percentile(x, 90):6.2f}  mean {x.mean():6.2f}")
print(f"{'cfg-5' if CFG5 else 'cfg-2'} x {B}" + ("" if CFG5 else f" radius {radii}") + f": {int(ok.sum())} non-empty workgroups of {nwg}; candidates per line p50 {np.median(n[ok]):.0f} max {n[ok].max():.0f}")
print("phase (kilocycles per workgroup)")
line("line load             0 -> 1", kc(0, 1))
line("row loads + LDS       1 -> 2", kc(1, 2))
line("staging barrier       2 -> 3", kc(2, 3))
walks = (t[:, 8:16] - t[:, 3:4]) / 1000.0
line("walk, fastest wave", walks.min(axis=1))
line("walk, mean wave", walks.mean(axis=1))
line("walk, slowest wave", walks.max(axis=1))
line("walk + barrier        3 -> 4", kc(3, 4))
line("write-out round 0     4 -> 5", kc(4, 5))
line("write-out rounds 1.. 5 -> 6", kc(5, 6))
line("workgroup life        0 -> 6", kc(0, 6))
