"""Host time of the reference harness's call form (test/test_time_numpy.py: forward(coords, center, channels, 1.0, 0.5, True,
out_grid=grid[i]) on the 10gs complex, 48^3):  python3 tools/latency_harness.py"""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import molvoxel_amd, bench_configs as bc

coords, center, types, features = bc.harness_inputs()
vox = molvoxel_amd.create_voxelizer(0.5, 48, library="hip")
dc, dcen = vox.asarray(coords, "coords"), vox.asarray(center, "center")
df = vox.asarray(features, "features")
dt = vox.asarray(types, "types")
grid = vox.get_empty_grid(10, batch_size=16)
for mode, ch in (("features", df), ("types", dt), ("single", None)):
    g = grid if ch is not None else vox.get_empty_grid(1, batch_size=16)
    for _ in range(200):
        vox.forward(dc, dcen, ch, 1.0, 0.5, True, out_grid=g[0])
    torch.cuda.synchronize()
    n = 4000
    t0 = time.perf_counter()
    for i in range(n):
        vox.forward(dc, dcen, ch, 1.0, 0.5, True, out_grid=g[i & 15])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host time per call {1e6 * (t1 - t0) / n:.1f} us, incl. drain {1e6 * (t2 - t0) / n:.1f} us")
pr = cProfile.Profile()
pr.enable()
for i in range(2000):
    vox.forward(dc, dcen, df, 1.0, 0.5, True, out_grid=grid[i & 15])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
