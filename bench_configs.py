#!/usr/bin/env python3
"""Kernel-level numbers for every BASELINE.json configuration (not the driver's contract: that is bench.py).

    python bench_configs.py [--steps 20]

For each config: batch size, voxelize-kernel time (HIP events), algorithmic GB/s, end-to-end ms per call.
cfg-1 and single-molecule rows also show the per-call latency of the drop-in `forward()` form.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def run(name, wl, batch_ids, steps, warmup=20):
    import torch

    import molvoxel_amd

    vox = molvoxel_amd.create_voxelizer(wl.resolution, wl.dimension, wl.radii_type, wl.density, library="hip",
                                        **({"sigma": wl.sigma} if wl.density == "gaussian" else {}))
    coords = [wl.coords[i] - wl.centers[i] for i in batch_ids]
    offsets = np.cumsum([0] + [c.shape[0] for c in coords]).astype(np.int64)
    d_coords = vox.asarray(np.concatenate(coords), "coords")
    chan = None
    if wl.mode == "features":
        chan = vox.asarray(np.concatenate([wl.channels[i] for i in batch_ids]), "features")
    elif wl.mode == "types":
        chan = torch.as_tensor(np.concatenate([wl.channels[i] for i in batch_ids]).astype(np.int32), device=vox.device)
    radii = wl.radii[batch_ids[0]]
    if not np.isscalar(radii):
        radii = vox.asarray(np.concatenate([wl.radii[i] for i in batch_ids]), "radii")
    B = len(batch_ids)
    out = vox.get_empty_grid(wl.num_channels, batch_size=B)
    if B == 1:  # the reference's own per-molecule form: forward(coords, center, channels, radii, out_grid=grid[i])
        step = lambda: vox.forward(d_coords, None, chan, radii, out_grid=out[0])
    else:
        step = lambda: vox.forward_batch(d_coords, offsets, None, chan, radii, num_channels=wl.num_channels, out_grid=out)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # kernel time in a pass of its own: the two timing events per launch cost several microseconds of a short call
    vox.set_profiling(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    k_ms = float(np.sum(vox.read_kernel_times_ms())) / steps  # voxelize launches of one call, summed
    alg = sum(wl.algorithmic_bytes(i) for i in batch_ids)
    return dict(config=name, batch=B, atoms=int(offsets[-1]), kernel_ms=k_ms, GBps=alg / (k_ms * 1e-3) / 1e9,
                ms_per_call=1e3 * el / steps, molecules_per_s=B * steps / el)


def harness_inputs():
    """The 10gs complex as test/test_time_numpy.py:24-50 voxelizes it, restated without RDKit: ligand + whole protein
    heavy atoms, centre = ligand centroid, channels = {C, N, O, S, other} x {ligand, protein} (the reference's
    getters add bond channels on top; atoms only here)."""
    pc = np.load(os.path.join(ROOT, "tests", "golden", "pointcloud_10gs.npz"))
    coords = np.concatenate([pc["ligand_xyz"], pc["protein_xyz"]])
    types = np.concatenate([pc["ligand_types"].astype(np.int64), pc["protein_types"].astype(np.int64) + 5])
    features = np.zeros((coords.shape[0], 10), dtype=np.float32)
    features[np.arange(coords.shape[0]), types] = 1.0
    return coords, pc["ligand_xyz"].mean(axis=0), types, features


def harness(voxelizer, batch_size=16, num_iteration=25, num_trial=5, log=print):
    """test/test_time_numpy.py:11-110 restated for any backend object: per-molecule `forward` calls with random
    translation 0.5 and random rotation into `out_grid=grid[i]`, 16 x 25 x 5, seconds per run for single / types /
    features. Returns {mode: seconds per run}."""
    coords, center, types, features = harness_inputs()
    coords, center = voxelizer.asarray(coords, "coords"), voxelizer.asarray(center, "center")
    types_, features_ = voxelizer.asarray(types, "types"), voxelizer.asarray(features, "features")
    grid = voxelizer.get_empty_grid(10, batch_size)
    single_grid = voxelizer.get_empty_grid(1, batch_size)

    def run_test(g, channels, tr=0.5, rot=True):
        for i in range(g.shape[0]):
            voxelizer.forward(coords, center, channels, 1.0, tr, rot, out_grid=g[i])
        return g

    def sync():
        if getattr(voxelizer, "LIB", "") == "HIP":
            import torch

            torch.cuda.synchronize()

    # sanity check of the reference harness (:52-69): reproducible, and types == one-hot features
    t = np.array(run_test(grid, types_, 0.0, False).tolist())
    f = np.array(run_test(grid, features_, 0.0, False).tolist())
    assert np.less(np.abs(t - t[0]), 1e-5).all() and np.less(np.abs(f - f[0]), 1e-5).all(), "REPRODUCTION FAIL"
    assert np.less(np.abs(t[0] - f[0]), 1e-5).all(), "REPRODUCTION FAIL"
    out = {}
    for mode, g, ch in (("single", single_grid, None), ("types", grid, types_), ("features", grid, features_)):
        sync()
        st = time.time()
        for _ in range(num_trial):
            for _ in range(num_iteration):
                run_test(g, ch)
        sync()
        out[mode] = (time.time() - st) / batch_size / num_iteration / num_trial
        log(f"{mode}: time per run {out[mode]:.3e} s")
    return out


def pcie_inclusive(steps=5):
    """cfg-2 through the numpy-in / numpy-out form of the boundary: host arrays up, the grid back over PCIe."""
    import molvoxel_amd
    from molvoxel_amd import workloads as W

    wl = W.cfg2(batch=8)
    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", output="numpy")
    coords = np.concatenate([wl.coords[i] for i in range(8)])
    feats = np.concatenate([wl.channels[i] for i in range(8)])
    offsets = np.arange(9, dtype=np.int64) * 4000
    vox.forward_batch(coords, offsets, None, feats, 1.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        vox.forward_batch(coords, offsets, None, feats, 1.0)
    el = time.perf_counter() - t0
    return dict(config="cfg2 x8, numpy in -> numpy out (PCIe inclusive)", molecules_per_s=8 * steps / el, ms_per_molecule=1e3 * el / steps / 8)


def pacing_rows(steps=40, warmup=60):
    """Guard rows for the pacing thresholds (mvx_tuning.h: write-out rounds paced from 49 152 workgroups = 96 cfg-2
    molecules, empty slabs held back beyond 4 096): cfg-2 at 16 / 64 / 96 / 256 molecules per call and cfg-4 x 128, kernel
    TB/s of algorithmic bytes. Re-run whenever the slab kernels change: a threshold that has drifted shows up as a dip at
    64 -> 96 or as cfg-4 falling below ~0.8 of peak (round 3: 0.715 / 0.785 / 0.80 / 0.80 of peak, cfg-4 0.85)."""
    from molvoxel_amd import workloads as W

    rows = []
    w2 = W.cfg2(batch=256)
    for b in (16, 64, 96, 256):
        r = run(f"pacing guard: cfg2 x{b}", w2, list(range(b)), steps, warmup)
        r["of_peak"] = r["GBps"] / 8000.0
        rows.append(r)
    w4 = W.cfg4(batch=128)
    r = run("pacing guard: cfg4 ligands x128", w4, list(range(128)), steps, warmup)
    r["of_peak"] = r["GBps"] / 8000.0
    rows.append(r)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--pacing", action="store_true", help="only the pacing guard rows (cfg-2 x 16 / 64 / 96 / 256, cfg-4 x 128)")
    ap.add_argument("--harness", action="store_true", help="also run the test_time_numpy.py loop (16 x 25 x 5) on the hip backend")
    args = ap.parse_args()
    from molvoxel_amd import workloads as W

    if args.pacing:
        for r in pacing_rows():
            print(json.dumps(r))
        return
    rows = []
    pc = np.load(os.path.join(ROOT, "tests", "golden", "pointcloud_10gs.npz"))
    w1 = W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"])
    rows.append(run("cfg1 ligand C=5 64^3 (single call)", w1, [0], args.steps))
    w2 = W.cfg2(batch=64)
    rows.append(run("cfg2 N=4000 C=32 64^3 (single call)", w2, [0], args.steps))
    rows.append(run("cfg2 N=4000 C=32 64^3 x64", w2, list(range(64)), args.steps))
    w3 = W.cfg3(batch=256)
    rows.append(run("cfg3 binary types 4ch 48^3 (single call)", w3, [0], args.steps))
    rows.append(run("cfg3 binary types 4ch 48^3 x256", w3, list(range(256)), args.steps))
    w4 = W.cfg4(batch=128)
    rows.append(run("cfg4 ligands C=16 64^3 x128", w4, list(range(128)), args.steps))
    w5 = W.cfg5(batch=4)
    rows.append(run("cfg5 N=10000 C=32 128^3 atom-wise (single call)", w5, [0], args.steps))
    rows.append(run("cfg5 N=10000 C=32 128^3 atom-wise x4", w5, [0, 1, 2, 3], args.steps))
    rows.append(pcie_inclusive())
    for r in rows:
        print(json.dumps(r))
    if args.harness:
        import molvoxel_amd

        res = harness(molvoxel_amd.create_voxelizer(0.5, 48, library="hip"), log=lambda *_: None)
        print(json.dumps(dict(config="test_time harness 10gs complex 48^3 (16 x 25 x 5), seconds per run", **res)))


if __name__ == "__main__":
    main()
