"""Step time of bench.py's workload without any timing event in the stream (python3 tools/step_time.py [batch] [overlap])."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench, molvoxel_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
overlap = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
wl, coords, feats = bench.make_batch(B, 0)
vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", overlap_prepass=overlap)
offsets = np.arange(B + 1, dtype=np.int64) * 4000
dc, df = vox.asarray(np.concatenate(coords), "coords"), vox.asarray(np.concatenate(feats), "features")
out = vox.get_empty_grid(32, batch_size=B)
for events in (False, True, False, True):
    vox.set_profiling(events)
    for _ in range(20):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    torch.cuda.synchronize()
    if events:
        vox.read_kernel_times_ms()
    t0 = time.perf_counter()
    for _ in range(40):
        vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 40
    print(f"overlap {overlap} kernel events {events}: {el * 1e3:.4f} ms per step, {B / el:.0f} molecules/s")
