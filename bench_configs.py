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


def run(name, wl, batch_ids, steps, warmup=3):
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
    step = lambda: vox.forward_batch(d_coords, offsets, None, chan, radii, num_channels=wl.num_channels, out_grid=out)
    for _ in range(warmup):
        step()
    vox.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k_ms = float(np.sum(vox.read_kernel_times_ms())) / steps  # voxelize launches of one call, summed
    alg = sum(wl.algorithmic_bytes(i) for i in batch_ids)
    return dict(config=name, batch=B, atoms=int(offsets[-1]), kernel_ms=k_ms, GBps=alg / (k_ms * 1e-3) / 1e9,
                ms_per_call=1e3 * el / steps, molecules_per_s=B * steps / el)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from molvoxel_amd import workloads as W

    rows = []
    pc = np.load(os.path.join(ROOT, "tests", "golden", "pointcloud_10gs.npz"))
    w1 = W.cfg1(pc["ligand_xyz"], pc["ligand_feat5"])
    rows.append(run("cfg1 ligand C=5 64^3 (single call)", w1, [0], args.steps))
    w2 = W.cfg2(batch=64)
    rows.append(run("cfg2 N=4000 C=32 64^3 (single call)", w2, [0], args.steps))
    rows.append(run("cfg2 N=4000 C=32 64^3 x64", w2, list(range(64)), args.steps))
    w3 = W.cfg3()
    rows.append(run("cfg3 binary types 4ch 48^3 (single call)", w3, [0], args.steps))
    w4 = W.cfg4(batch=128)
    rows.append(run("cfg4 ligands C=16 64^3 x128", w4, list(range(128)), args.steps))
    w5 = W.cfg5(batch=4)
    rows.append(run("cfg5 N=10000 C=32 128^3 atom-wise (single call)", w5, [0], args.steps))
    rows.append(run("cfg5 N=10000 C=32 128^3 atom-wise x4", w5, [0, 1, 2, 3], args.steps))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
