"""Per-molecule `forward()` calls of one BASELINE configuration, for rocprofv3 --kernel-trace --stats and host timing.

    python3 tools/single_calls.py cfg2 [calls]      (cfg1 | cfg2 | cfg3 | cfg5 | harness)
Prints host-side us per call (back-to-back, drained at the end).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
from molvoxel_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if len(sys.argv) > 3:  # A/B builds: another libmvx_hip.so (path relative to the repo root)
    from molvoxel_amd.voxelizer.hip import _lib as _l

    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), sys.argv[3])
pc = np.load(os.path.join(ROOT, "tests", "golden", "pointcloud_10gs.npz"))
tr, rot = 0.0, False
if name == "harness":  # test/test_time_numpy.py: 10gs complex, 48^3, C = 10, random transform per call
    import bench_configs

    xyz, center, types, feats = bench_configs.harness_inputs()
    vox = molvoxel_amd.create_voxelizer(0.5, 48, library="hip")
    coords, cen = vox.asarray(xyz, "coords"), vox.asarray(center, "center")
    chan, radii, C_ = vox.asarray(feats, "features"), 1.0, 10
    if os.environ.get("MODE") == "types":  # MODE=single|types python3 tools/single_calls.py harness
        chan = vox.asarray(types, "types")
    elif os.environ.get("MODE") == "single":
        chan, C_ = None, 1
    tr, rot = 0.5, True
else:
    wl = {"cfg1": lambda: W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"]), "cfg2": W.cfg2, "cfg3": W.cfg3, "cfg5": W.cfg5}[name]()
    vox = molvoxel_amd.create_voxelizer(wl.resolution, wl.dimension, wl.radii_type, wl.density, library="hip",
                                        **({"sigma": wl.sigma} if wl.density == "gaussian" else {}))
    coords = vox.asarray(wl.coords[0] - wl.centers[0], "coords")
    cen = None
    chan = vox.asarray(wl.channels[0], wl.mode)
    radii = wl.radii[0] if np.isscalar(wl.radii[0]) else vox.asarray(wl.radii[0], "radii")
    C_ = wl.num_channels
for opt in ("direct",):  # route switch: DIRECT=0 python3 tools/single_calls.py cfg2
    if os.environ.get(opt.upper()) is not None:
        vox.debug_option(opt, int(os.environ[opt.upper()]))
if os.environ.get("MVX_DBG"):  # diagnostic builds: run-time ablations of the direct kernel
    vox.debug_option("dbg", int(os.environ["MVX_DBG"]))
if os.environ.get("MVX_MAX_CT"):  # narrower chunks: more, lighter workgroups per call
    vox.debug_option("max_ct", int(os.environ["MVX_MAX_CT"]))
if os.environ.get("MVX_DENSE_GRID"):
    vox.debug_option("dense_grid", int(os.environ["MVX_DENSE_GRID"]))
grid = vox.get_empty_grid(C_)
for _ in range(20):
    vox.forward(coords, cen, chan, radii, tr, rot, out_grid=grid)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    vox.forward(coords, cen, chan, radii, tr, rot, out_grid=grid)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
vox.set_profiling(True)
for _ in range(200):
    vox.forward(coords, cen, chan, radii, tr, rot, out_grid=grid)
torch.cuda.synchronize()
kt = np.sort(np.array(vox.read_kernel_times_ms())) * 1e3
vox.set_profiling(False)
print(f"{name}: main kernel (HIP events) min {kt[0]:.1f} p50 {kt[len(kt) // 2]:.1f} us")
print(f"{name}: host {1e6 * (t1 - t0) / calls:.1f} us/call, with drain {1e6 * (t2 - t0) / calls:.1f} us/call over {calls} calls")
