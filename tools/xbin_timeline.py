"""Per-phase timeline of xbin_kernel for one cfg-5 molecule (N = 10 000, 128^3) from a -DMVX_DIAG build.

    python3 tools/xbin_timeline.py [lib]
Stamps by thread 0 of every block (s_memtime, shader cycles):
  0 start | 1 pass A done (x-list in LDS) | 2 pass B over the LDS copy (wave 0's last group) | 3 read-back tail |
  4 wave 0's lines stored | 5 block barrier | 7 = x-list length
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molvoxel_amd.voxelizer.hip import _lib as _l

_l.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "molvoxel_amd/csrc/ab/libmvx_diag.so")
_l.SIGNATURES["mvx_debug_read_diag"] = (C.c_int, [_l.Handle, C.c_void_p, C.c_int64])
import molvoxel_amd
from molvoxel_amd import workloads as W

wl = W.cfg5()
vox = molvoxel_amd.create_voxelizer(0.5, 128, "atom-wise", "gaussian", library="hip", sigma=1.0)
c = vox.asarray(wl.coords[0], "coords")
f = vox.asarray(wl.channels[0], "features")
r = vox.asarray(wl.radii[0], "radii")
g = vox.get_empty_grid(32)
for _ in range(30):
    vox.forward(c, None, f, r, out_grid=g)
torch.cuda.synchronize()
nblk = 4096
vox.debug_option("xb_stamps", nblk)
vox.forward(c, None, f, r, out_grid=g)
buf = np.zeros((nblk, 8), dtype=np.uint64)
_l.check(vox._lib.mvx_debug_read_diag(vox._handle, buf.ctypes.data, buf.nbytes))
vox.debug_option("xb_stamps", 0)
t = buf.astype(np.float64)
t = t[t[:, 0] > 0]
print(f"{len(t)} blocks; x-list length p50 {np.median(t[:, 7]):.0f} max {t[:, 7].max():.0f}")
span = (t[:, 5].max() - t[:, 0].min()) / 1000.0
print(f"first start -> last end: {span:.1f} kilocycles (only comparable within an XCD's clock domain; indicative)")


def line(name, x):
    print(f"{name:34s} p10 {np.percentile(x, 10):6.2f}  p50 {np.percentile(x, 50):6.2f}  p90 {np.percentile(x, 90):6.2f}  mean {x.mean():6.2f}")


kc = lambda a, b: (t[:, b] - t[:, a]) / 1000.0
print("phase (kilocycles per block)")
line("pass A                0 -> 1", kc(0, 1))
line("  loads + ballots     0 -> 6", kc(0, 6))
line("  rest of pass A      6 -> 1", kc(6, 1))
line("pass B, LDS part      1 -> 2", kc(1, 2))
line("pass B, read-back     2 -> 3", kc(2, 3))
line("line stores           3 -> 4", kc(3, 4))
line("barrier               4 -> 5", kc(4, 5))
line("block life            0 -> 5", kc(0, 5))
line("start offset from first block", (t[:, 0] - t[:, 0].min()) / 1000.0)
