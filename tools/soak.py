"""Long randomised parity run beyond the committed fuzz seeds.

    python3 tools/soak.py FIRST COUNT            single-molecule configurations (tests/test_hip_fuzz._draw)
    python3 tools/soak.py batches FIRST COUNT    ragged batches (tests/test_hip_fuzz.test_random_batches with other seeds)
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
    if precision == 64 and case["C"] > 32:
        continue
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
    if (seed - first) % 200 == 199:
        print(f"{seed - first + 1} cases, {len(bad)} bad, {time.time() - t0:.0f}s", flush=True)
print("done", count, "cases; bad seeds:", bad)
