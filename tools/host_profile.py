"""Where the host time of a per-molecule forward() call goes (cProfile over N calls; python3 tools/host_profile.py cfg3 [calls])."""
import cProfile, pstats, sys, os, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
from molvoxel_amd import workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
wl = {"cfg2": W.cfg2, "cfg3": W.cfg3}[name]()
vox = molvoxel_amd.create_voxelizer(wl.resolution, wl.dimension, wl.radii_type, wl.density, library="hip",
                                    **({"sigma": wl.sigma} if wl.density == "gaussian" else {}))
coords = vox.asarray(wl.coords[0], "coords")
chan = vox.asarray(wl.channels[0], wl.mode)
grid = vox.get_empty_grid(wl.num_channels)
for _ in range(50): vox.forward(coords, None, chan, 1.0, out_grid=grid)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(calls): vox.forward(coords, None, chan, 1.0, out_grid=grid)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(f"{name}: {calls} calls")
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:4000])
