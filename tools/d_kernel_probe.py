"""Kernel time (HIP events) next to the whole call for grids of different row lengths at cfg-2 density, ~0.5 GB of grid per call:
    python3 tools/d_kernel_probe.py [D ...]     (NW=<n> forces the slab plan's waves per slab)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
rng = np.random.default_rng(0)
C = int(os.environ.get("CHANNELS", "32"))
for D in [int(a) for a in sys.argv[1:]] or [64, 72, 80, 96, 112, 128]:
    B = max(4, int(round(64 * (64.0 / D) ** 3)))
    N = int(os.environ["ATOMS"]) if os.environ.get("ATOMS") else max(8, int(4000 * (D / 64.0) ** 3))  # (ATOMS: a fixed count instead of cfg-2 density)
    vox = molvoxel_amd.create_voxelizer(0.5, D, library="hip")
    if os.environ.get("NW"):
        vox.debug_option("nw", int(os.environ["NW"]))
    W = 0.5 * (D - 1)
    coords = vox.asarray(rng.uniform(-W / 2, W / 2, (B * N, 3)), "coords")
    chan = vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
    off = np.arange(B + 1, dtype=np.int64) * N
    out = vox.get_empty_grid(C, batch_size=B)
    call = lambda: vox.forward_batch(coords, off, None, chan, 1.0, out_grid=out)
    for _ in range(25):
        call()
    torch.cuda.synchronize()
    vox.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(30):
        call()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 30
    k = np.median(np.array(vox.read_kernel_times_ms()))
    vox.set_profiling(False)
    nbytes = 4.0 * B * C * D**3
    print(f"D = {D:3d} C = {C} x {B:3d} ({N} atoms each) NW {os.environ.get('NW', 'plan')}: call {el*1e3:.3f} ms ({nbytes/el/1e12:.2f} TB/s of grid bytes), voxelize kernel {k:.3f} ms ({nbytes/k/1e9:.2f} TB/s)")
