"""Per-phase timeline of voxelize_pair_kernel (the per-molecule launch) from a -DMVX_DIAG build (tools/ab_build.sh diag "-DMVX_DIAG").

    python3 tools/direct_timeline.py cfg1|cfg2|harness [lib]
Stamps (s_memtime, 100 MHz constant clock -> 10 ns ticks... printed in us) per workgroup:
  0 start | 1 scan done (+ barrier, prefix) | 2 rows staged (this wave) | 3 barrier passed | 4 walk done (wave 0) |
  5 all rounds done | 6 write-out issued
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
import molvoxel_amd
from molvoxel_amd import workloads as W

name = sys.argv[1]
pc = np.load(os.path.join(ROOT, "tests", "golden", "pointcloud_10gs.npz"))
tr, rot, cen = 0.0, False, None
if name == "harness":
    import bench_configs

    xyz, center, types, feats = bench_configs.harness_inputs()
    vox = molvoxel_amd.create_voxelizer(0.5, 48, library="hip")
    coords, cen = vox.asarray(xyz, "coords"), vox.asarray(center, "center")
    chan, radii, C_, D = vox.asarray(feats, "features"), 1.0, 10, 48
    tr, rot = 0.5, True
else:
    wl = {"cfg1": lambda: W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"]), "cfg2": W.cfg2, "cfg3": W.cfg3}[name]()
    vox = molvoxel_amd.create_voxelizer(wl.resolution, wl.dimension, wl.radii_type, wl.density, library="hip", sigma=wl.sigma)
    coords = vox.asarray(wl.coords[0] - wl.centers[0], "coords")
    chan = vox.asarray(wl.channels[0], wl.mode)
    radii, C_, D = wl.radii[0], wl.num_channels, wl.dimension
grid = vox.get_empty_grid(C_)
for _ in range(20):
    vox.forward(coords, cen, chan, radii, tr, rot, out_grid=grid)
torch.cuda.synchronize()
nwg = ((D + 3) // 4) * ((D + 3) // 4)  # voxelize_pair_kernel: one workgroup per pair of x-slabs (D % 4 == 0)
buf = np.zeros((nwg, 16), dtype=np.uint64)
_l.check(vox._lib.mvx_debug_read_records(vox._handle, buf.ctypes.data, 2 * nwg, 0))  # (64-byte records: 16 stamps = 2)
t = buf.astype(np.float64)
t0 = t[:, 0].min()
rel = (t - t0) / 1000.0  # s_memtime ticks = shader cycles; printed in kcycles (~0.45 us each at 2.2 GHz)
has = t[:, 2] > 0
print(f"{name}: {nwg} workgroups, {int(has.sum())} with candidates; (per-workgroup deltas in kcycles; absolute times are not comparable across XCDs)")
def stats(x):
    return f"min {x.min():6.2f}  p50 {np.median(x):6.2f}  max {x.max():6.2f}"
print("start           ", stats(rel[:, 0]))
print("scan (0->1)     ", stats(rel[:, 1] - rel[:, 0]))
print("  last wave starts  ", stats(rel[:, 13] - rel[:, 0]))
print("  prologue    0->8  ", stats(rel[:, 8] - rel[:, 0]))
sc = t[:, 9] > 0
if sc.any():
    print("  loads issued 8->9 ", stats((rel[:, 9] - rel[:, 8])[sc]))
    print("  first data  9->10 ", stats((rel[:, 10] - rel[:, 9])[sc]))
    print("  tests (last wave) 10->11", stats((rel[:, 11] - rel[:, 10])[sc]))
    print("  barrier+prefix 11->1    ", stats((rel[:, 1] - rel[:, 11])[sc]))
e = ~has
if e.any():
    print("empty: fill 5->6", stats((rel[:, 6] - rel[:, 5])[e]), " end", stats(rel[:, 6][e]))
if has.any():
    h = has
    print("stage (1->2)    ", stats((rel[:, 2] - rel[:, 1])[h]))
    print("  coords in 1->7", stats((rel[:, 7] - rel[:, 1])[h]))
    print("  math+LDS  7->2", stats((rel[:, 2] - rel[:, 7])[h]))
    print("  last wave staged 1->12", stats((rel[:, 12] - rel[:, 1])[h]))
    print("barrier (2->3)  ", stats((rel[:, 3] - rel[:, 2])[h]))
    print("walk (3->4)     ", stats((rel[:, 4] - rel[:, 3])[h]))
    print("write (5->6)    ", stats((rel[:, 6] - rel[:, 5])[h]))
    print("end             ", stats(rel[:, 6][h]))
