"""Grids whose rows are not whole 16-byte quads (D % 4 != 0), and unaligned grid base addresses: whole-call rate next to the
neighbouring aligned sizes (cfg-2 density: 4000 atoms at D = 64, scaled by volume; 64 molecules per call).
    python3 tools/odd_d_probe.py [lib]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), sys.argv[1])
import molvoxel_amd

B = int(os.environ.get("BATCH", "64"))  # molecules per call
rng = np.random.default_rng(0)


def run(D, C=32, shift=0, empty=False, nw=0):
    N = 8 if empty else max(8, int(4000 * (D / 64.0) ** 3))
    vox = molvoxel_amd.create_voxelizer(0.5, D, library="hip")
    if nw:
        vox.debug_option("nw", nw)
    if os.environ.get("DIRECT"):  # 0: the binned pipeline, 1: the one-launch route (where it applies)
        vox.debug_option("direct", int(os.environ["DIRECT"]))
    W = 0.5 * (D - 1)
    coords = vox.asarray(rng.uniform(-W / 2, W / 2, (B * N, 3)), "coords")
    chan = vox.asarray(rng.random((B * N, C)).astype(np.float32), "features")
    off = np.arange(B + 1, dtype=np.int64) * N
    n = B * C * D**3
    flat = torch.empty(n + 4, dtype=torch.float32, device="cuda")
    out = flat[shift:shift + n].view(B, C, D, D, D)  # shift floats off the allocation's 16-B alignment
    call = lambda: vox.forward_batch(coords, off, None, chan, 1.0, out_grid=out)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20
    print(f"D = {D:3d} C = {C:2d} x {B:4d} nw {nw:2d} base + {4 * shift:2d} B {'(8 atoms)' if empty else '':9s} {el * 1e3:8.3f} ms/call  {4 * n / el / 1e12:5.2f} TB/s of grid bytes")


if os.environ.get("NW_SWEEP"):  # long rows cut into chunks of 8 sub-tiles (nw 0 = the plan) against balanced chunks
    for D, C, nws in ((65, 32, (0, 5)), (66, 32, (0, 5)), (72, 32, (0, 5)), (80, 32, (0, 5)), (96, 16, (0, 6)), (100, 8, (0, 7)),
                      (101, 8, (0, 7)), (104, 8, (0, 7)), (120, 8, (0, 8, 5)), (128, 8, (0, 6))):
        for nw in nws:
            run(D, C, nw=nw)
    sys.exit(0)
if os.environ.get("SIZE_SWEEP"):  # aligned sizes up to 64, ~0.5 GB of grids per call
    for D in (8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64):
        for C in (32, 8):
            B = max(8, min(4096, (1 << 29) // (C * D**3 * 4)))
            run(D, C)
    sys.exit(0)
if os.environ.get("ONE_SWEEP"):  # one molecule per call, rows of 65 ... 96 voxels (BATCH=1 DIRECT=0|1)
    for D in (64, 68, 72, 76, 80, 88, 96):
        run(D, 32)
        run(D, 32, empty=True)  # 8 atoms: a ligand-sized call
        run(D, 8)
    sys.exit(0)
if os.environ.get("ROUTE_SWEEP"):  # one molecule per call up to D = 64: is the route rule's choice the faster one? (BATCH=1, DIRECT unset / 0 / 1)
    for D in (24, 32, 48, 49, 50, 63, 64):
        for C in (1, 8, 32, 64):
            run(D, C)
            run(D, C, empty=True)
    sys.exit(0)
if os.environ.get("SMALL_BATCH_SWEEP"):  # a few molecules per call: where the one-launch route should hand over (DIRECT unset / 0 / 1)
    for D, C in ((64, 32), (64, 8), (48, 16), (32, 32)):
        for B in (2, 3, 4, 6, 8, 16):
            run(D, C)
            run(D, C, empty=True)
    sys.exit(0)
if os.environ.get("PACE_SWEEP"):  # aligned sizes with 32-channel chunks, ~2 GB of grids per call: where the round pacing applies
    for D in (24, 32, 40, 48, 56, 64):
        B = max(8, (1 << 31) // (32 * D**3 * 4))
        run(D, 32)
    sys.exit(0)
if os.environ.get("LONG_SWEEP"):  # rows of more than 128 voxels: chunks of 8 sub-tiles (nw 0) against fewer, longer chunks
    for D, C, nws in ((136, 32, (0, 9)), (144, 32, (0, 9, 16)), (152, 32, (0, 10)), (160, 32, (0, 10, 16)), (168, 32, (0, 11)),
                      (192, 16, (0, 12, 16)), (200, 16, (0, 13)), (256, 8, (0, 16))):
        for nw in nws:
            run(D, C, nw=nw)
    sys.exit(0)
if os.environ.get("ROW_SWEEP"):  # 64 < D <= 128: chunks of 8 sub-tiles (nw 0 = the plan) against whole rows in one slab
    for D, C in ((72, 32), (80, 32), (88, 32), (96, 32), (104, 32), (112, 32), (120, 32), (128, 32), (65, 32), (100, 8), (72, 8), (72, 16), (88, 4), (120, 16)):
        for nw in (0, 8, (D + 7) // 8):
            run(D, C, nw=nw)
    sys.exit(0)
for D in (48, 49, 50, 51, 52, 63, 64, 65, 66):
    run(D)
run(64, shift=1)
run(50, empty=True)
run(100, C=8)
run(101, C=8)
run(102, C=8)
