"""Channel-wise features at cfg-2 geometry (64 molecules per call): python3 tools/chanwise_probe.py <distinct radii> [calls]
(for rocprofv3 --kernel-trace --stats: which launch the time goes to)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molvoxel_amd
B, N, C = 64, 4000, 32
k = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(0)
W = 0.5 * 63
vox = molvoxel_amd.create_voxelizer(0.5, 64, "channel-wise", "gaussian", library="hip")
coords = vox.asarray(rng.uniform(-W / 2, W / 2, (B * N, 3)), "coords")
feats = vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
radii = vox.asarray(np.resize(np.linspace(0.85, 1.15, k).astype(np.float32), C), "radii")
off = np.arange(B + 1, dtype=np.int64) * N
out = vox.get_empty_grid(C, batch_size=B)
for _ in range(5): vox.forward_batch(coords, off, None, feats, radii, out_grid=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(calls): vox.forward_batch(coords, off, None, feats, radii, out_grid=out)
torch.cuda.synchronize()
print(f"{k} distinct radii: {(time.perf_counter() - t0) / calls * 1e3:.3f} ms per call")
