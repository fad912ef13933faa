import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import molvoxel_amd
from molvoxel_amd import workloads as W
wl = W.cfg5()
for ct, direct in ((32, 0), (32, 1)):
    vox = molvoxel_amd.create_voxelizer(0.5, 128, "atom-wise", "gaussian", library="hip", sigma=1.0)
    vox.debug_option("max_ct", ct); vox.debug_option("direct", direct)
    c = vox.asarray(wl.coords[0], "coords"); f = vox.asarray(wl.channels[0], "features"); r = vox.asarray(wl.radii[0], "radii")
    g = vox.get_empty_grid(32)
    for _ in range(30): vox.forward(c, None, f, r, out_grid=g)
    torch.cuda.synchronize()
    vox.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(200): vox.forward(c, None, f, r, out_grid=g)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 200
    k = np.array(vox.read_kernel_times_ms()) * 1e3
    per = len(k) // 200
    print(f"max_ct {ct} direct {direct}: {el*1e6:.1f} us/call, voxelize launches/call {per}, kernel sum {k.sum()/200:.1f} us, {270.0e6/el/1e12:.2f} TB/s end to end")
