#!/usr/bin/env python3
"""Headline benchmark: forward_features, C=32, 64^3, N=4000 (BASELINE.json configs[1]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one pass of the hot path over one batch of `--batch` synthetic cfg-2 molecules per GPU
(one mvx_forward_features_batch call: prep kernel + voxelize kernel), inputs already resident in
HBM, outputs left in HBM. value = molecules/s over all ranks (weak scaling: per-GPU batch fixed;
molecules are independent, so ranks share nothing and there is no collective on the data path).
Rank 0 prints ONE JSON line with `roofline` (voxelize kernel, HIP events on the launch stream,
algorithmic bytes of SURVEY.md §8d) and `cpu_baseline` (numpy port of the reference algorithm timed
on this host, N=1 only).
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290


def make_batch(batch: int, rank: int):
    from molvoxel_amd import workloads as W

    # molecule j of the job is cfg-2 with seed 1000*j; rank r owns molecules [r*batch, (r+1)*batch)
    wl = W.cfg2(batch=1)  # template (geometry)
    coords, feats = [], []
    for j in range(rank * batch, (rank + 1) * batch):
        rng = np.random.default_rng(0 + 1000 * j)
        Wd = 0.5 * 63
        coords.append(rng.uniform(-Wd / 2, Wd / 2, (4000, 3)))
        feats.append(rng.random((4000, 32)).astype(np.float32))
    return wl, coords, feats


def cpu_baseline(budget_s: float):
    """numpy port of the reference block algorithm (oracle/numpy_port.py) on one cfg-2 molecule."""
    from oracle import c_oracle, numpy_port

    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    Wd = 0.5 * 63
    xyz = rng.uniform(-Wd / 2, Wd / 2, (4000, 3))
    feat = rng.random((4000, 32)).astype(np.float32)
    spec = numpy_port.GridSpec(0.5, 64)
    out = np.empty((32, 64, 64, 64), np.float32)
    numpy_port.voxelize(spec, xyz, feat, 1.0, out=out)  # warm (block grids cached)
    n, t0 = 0, time.perf_counter()
    while True:
        numpy_port.voxelize(spec, xyz, feat, 1.0, out=out)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 400:
            break
    port = dict(value=n / el, unit="molecules/s", cores=int(blas_threads), kind="port",
                sample=f"{n} calls of oracle/numpy_port.voxelize (numpy {np.__version__} + scipy cdist/BLAS, the "
                       f"reference's block algorithm) on the seed-0 cfg-2 molecule in {el:.1f} s; "
                       f"os.cpu_count()={os.cpu_count()}, BLAS threads={blas_threads}")
    # all-core C/OpenMP restatement of the same rule (fairest CPU number), reported alongside
    c_oracle.voxelize(xyz, feat, 1.0, dimension=64, out=out)
    m, t0 = 0, time.perf_counter()
    while True:
        c_oracle.voxelize(xyz, feat, 1.0, dimension=64, out=out)
        m += 1
        el2 = time.perf_counter() - t0
        if el2 >= budget_s / 3 or m >= 2000:
            break
    port["openmp_port"] = dict(value=m / el2, unit="molecules/s", cores=c_oracle.num_threads(),
                               sample=f"{m} calls of oracle/mvx_oracle.c (OpenMP) in {el2:.1f} s")
    return port


def load_pmc_traffic(molecules_per_launch: int):
    """HBM bytes per voxelize launch from the committed rocprofv3 --pmc summary, if it matches this config."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get("workload") == "cfg2" and int(d.get("molecules_per_launch", -1)) == molecules_per_launch:
            return float(d["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="cfg-2 molecules per GPU per step (8.6 GB of grids at 256)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    ndev = torch.cuda.device_count()
    # one GPU per rank is the real layout; with fewer GPUs than ranks (rehearsal on a 1-GPU box) ranks share devices
    # and the barrier runs over gloo, because RCCL refuses two ranks on one device
    dev_index = local_rank if ndev >= world else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ndev >= world:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import molvoxel_amd

    B = args.batch
    wl, coords, feats = make_batch(B, rank)
    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", device=dev_index)
    offsets = np.arange(B + 1, dtype=np.int64) * 4000
    d_coords = vox.asarray(np.concatenate(coords), "coords")
    d_feats = vox.asarray(np.concatenate(feats), "features")
    out = vox.get_empty_grid(32, batch_size=B)

    def step():
        vox.forward_batch(d_coords, offsets, None, d_feats, 1.0, out_grid=out)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    vox.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = vox.read_kernel_times_ms()
    vox.set_profiling(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # the library cuts a step's batch into equal chunks of molecules (pre-pass of chunk k+1 overlaps the
        # voxelize launch of chunk k): per launch = B / launches_per_step molecules x (4*C*D^3 + N*(24 + 4*C + 4))
        lps = max(1, len(kernel_ms) // args.steps)
        alg_bytes = (B // lps) * wl.algorithmic_bytes(0)
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "molecules/sec + achieved HBM GB/s, forward_features C=32 64^3 N=4000",
            "value": args.gpus * B * args.steps / elapsed,
            "unit": "molecules/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: forward_features, gaussian sigma=0.5, scalar radius 1.0, C=32, 64^3, N=4000 atoms/molecule",
                "molecules_per_gpu_per_step": B,
                "inputs": "HBM-resident (torch CUDA tensors), outputs left in HBM",
                "geometry_dtype": "f64",
                "parallelism": f"{args.gpus} independent ranks, molecules sharded, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "voxelize_kernel<32,gauss>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "kernel_ms_avg": k_ms,
                "kernel_launches_timed": len(kernel_ms),
                "launches_per_step": lps,
                "molecules_per_launch": B // lps,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic": load_pmc_traffic(B // lps),
            },
        }
        if args.gpus == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
