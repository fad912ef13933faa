"""Long randomised parity run beyond the committed fuzz seeds.

    python3 tools/soak.py FIRST COUNT            single-molecule configurations (tests/test_hip_fuzz._draw)
    python3 tools/soak.py batches FIRST COUNT    ragged batches (tests/test_hip_fuzz.test_random_batches with other seeds)
    python3 tools/soak.py routes FIRST COUNT     direct kernel against binned pipeline, bit for bit, on inputs that stress the
                                                 direct kernel's float32 candidate scan: far-away centres (|c| up to 1e5),
                                                 random rotations / translations, every radii kind and operator
"""
import importlib.util
import sys
import time

import numpy as np

sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz", "tests/test_hip_fuzz.py")
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
import molvoxel_amd as mv
import os

if os.environ.get("SOAK_DIMS"):  # e.g. SOAK_DIMS=65,66,72,88,100,101,127: grid sizes outside the committed fuzz list
    fz.DIMS = [int(x) for x in os.environ["SOAK_DIMS"].split(",")]

if sys.argv[1] == "routes":
    import torch

    first, count = int(sys.argv[2]), int(sys.argv[3])
    bad, t0 = [], time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(900_000 + seed)
        D = int(rng.choice([16, 24, 33, 48, 64]))
        res = float(rng.choice([0.4, 0.5, 1.0]))
        W = res * (D - 1)
        n = int(rng.choice([1, 7, 48, 49, 64, 65, 300, 2000, 4096, 4100, 9000]))
        scale = float(rng.choice([0.0, 10.0, 1e3, 1e4, 1e5]))
        center = rng.uniform(-1, 1, 3) * scale
        xyz = rng.uniform(-W / 2 - 2, W / 2 + 2, (n, 3)) + center
        mode = str(rng.choice(["features", "types", "single"]))
        radii_type = str(rng.choice(["scalar", "atom-wise"] + ([] if mode == "single" else ["channel-wise"])))
        density = str(rng.choice(["gaussian", "binary"]))
        C_ = 1 if mode == "single" else int(rng.choice([1, 3, 8, 16, 32, 40]))
        bd = rng.choice([None, None, 4, 5, D])
        extra = {} if bd is None else {"blockdim": int(bd)}
        v = mv.create_voxelizer(res, D, radii_type, density, "hip", sigma=0.6, **extra)
        chan = None if mode == "single" else (rng.random((n, C_)).astype(np.float32) if mode == "features" else rng.integers(0, C_, n))
        if mode == "types":
            chan[0] = C_ - 1  # max(types) + 1 == C, what channel-wise radii must match
        radii = {"scalar": 1.3 * res / 0.5, "atom-wise": (rng.uniform(0.8, 2.0, n) * res / 0.5).astype(np.float32),
                 "channel-wise": (rng.uniform(0.8, 2.0, C_) * res / 0.5).astype(np.float32)}[radii_type]
        dx, dc = v.asarray(xyz, "coords"), v.asarray(center, "center")
        dch = None if chan is None else v.asarray(chan, mode)
        dr = radii if np.isscalar(radii) else v.asarray(radii, "radii")
        tr, rot = float(rng.choice([0.0, 0.5, 3.0])), bool(rng.random() < 0.7)
        outs = []
        for route in (0, 1):
            v.debug_option("direct", route)
            np.random.seed(seed)
            outs.append(v.forward(dx, dc, dch, dr, tr, rot).clone())
        if not torch.equal(outs[0], outs[1]):
            bad.append(seed)
            print("MISMATCH routes seed", seed, dict(D=D, n=n, scale=scale, mode=mode, radii_type=radii_type, density=density, C=C_, bd=bd, tr=tr, rot=rot),
                  int((outs[0] != outs[1]).sum()), flush=True)
        if (seed - first) % 200 == 199:
            print(f"{seed - first + 1} route cases, {len(bad)} bad, {time.time() - t0:.0f}s", flush=True)
    print("done", count, "route cases; bad seeds:", bad)
    sys.exit(0)

if sys.argv[1] == "batches":
    first, count = int(sys.argv[2]), int(sys.argv[3])
    bad, t0 = [], time.time()
    body = fz.test_random_batches.__wrapped__ if hasattr(fz.test_random_batches, "__wrapped__") else fz.test_random_batches
    for seed in range(first, first + count):
        try:
            body(seed)
        except Exception as e:  # noqa: BLE001
            bad.append(seed)
            print("MISMATCH batch seed", seed, repr(e)[:300], flush=True)
        if (seed - first) % 50 == 49:
            print(f"{seed - first + 1} batches, {len(bad)} bad, {time.time() - t0:.0f}s", flush=True)
    print("done", count, "batches; bad seeds:", bad)
    sys.exit(0)

first, count = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = [], time.time()
for seed in range(first, first + count):
    case = fz._draw(seed)
    if case["N"] == 0 and case["mode"] == "types":
        continue
    precision = 64 if seed % 8 == 7 else 32
    try:
        out, moved = fz._run(mv, case, precision)
        ref = fz._reference(case, moved, precision)
        ok = out.shape == ref.shape and np.array_equal(out != 0, ref != 0)
        if ok:
            if case["density"] == "binary" and case["mode"] != "features":
                ok = np.array_equal(out, ref)
            else:
                tol = (5e-6 if precision == 32 else 1e-12) * max(1.0, float(np.abs(ref).max()))
                ok = float(np.abs(out - ref).max()) <= tol
    except Exception as e:  # noqa: BLE001
        ok = False
        print("seed", seed, "raised", repr(e)[:200])
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, {k: v for k, v in case.items() if k not in ("xyz", "chan", "radii", "center")}, flush=True)
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} cases, {len(bad)} bad, {time.time() - t0:.0f}s", flush=True)
print("done", count, "cases; bad seeds:", bad)
