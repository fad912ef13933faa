import os, sys, numpy as np, ctypes as C
os.environ["MVX_STAMPS"] = "1"
sys.path.insert(0, ".")
import molvoxel_amd
from molvoxel_amd.voxelizer.hip import _lib
from bench import make_batch
B = 64
wl, coords, feats = make_batch(B, 0)
vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip")
offsets = np.arange(B + 1, dtype=np.int64) * 4000
dc = vox.asarray(np.concatenate(coords), "coords"); df = vox.asarray(np.concatenate(feats), "features")
out = vox.get_empty_grid(32, batch_size=B)
for _ in range(3):
    vox.forward_batch(dc, offsets, None, df, 1.0, out_grid=out)
nb = B * 1024
buf = np.zeros((nb, 8), dtype=np.uint64)
n = C.c_int64(0)
_lib.check(vox._lib.mvx_debug_read_stamps(vox._handle, buf.ctypes.data, nb, C.byref(n)))
s = buf[: n.value].astype(np.int64)
ok = (s[:, [0, 1, 2, 3, 4, 5, 7]] > 0).all(axis=1)
s = s[ok]
print("blocks", n.value, "with candidates", len(s))
for a, b_, name in [(0,1,"start->loads issued"),(1,2,"->scan done (barrier)"),(2,3,"->staged (barrier)"),(3,4,"->walk done (thread 0)"),(4,5,"->tile0 ready (2 barriers)"),(5,7,"->end"),(0,7,"total")]:
    d = s[:, b_] - s[:, a]
    print(f"{name:30s} mean {d.mean():9.0f} p50 {np.median(d):9.0f} p90 {np.percentile(d,90):9.0f}")
