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


def parity_spot(out, coords, feats, picks, radius=1.0, dimension=64, sigma=0.5):
    """Post-timing spot check (outside the timed region): the grids of the molecules `picks` of the batch that was
    just timed against the CPU oracle: membership identical, |d| <= 5e-6 * max(1, |ref|) per voxel
    (tests/tolerance.py). Returns "ok" or "FAIL: ..."."""
    from oracle import c_oracle

    for b in picks:
        ref = c_oracle.voxelize(coords[b], feats[b], radius, dimension=dimension, sigma=sigma)
        got = out[b].cpu().numpy()
        bad = int(np.not_equal(got != 0, ref != 0).sum())
        if bad:
            return f"FAIL: molecule {b}: membership differs in {bad} voxels"
        ex = float((np.abs(got - ref) / (5e-6 * np.maximum(1.0, np.abs(ref)))).max())
        if ex > 1.0:
            return f"FAIL: molecule {b}: error {ex:.3g} x the tolerance"
    return "ok"


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


def make_cfg4_shard(total: int, rank: int, world: int):
    """cfg-4 (BASELINE.json configs[3]): `total` ligands (40-60 atoms, C = 16) cut into contiguous shards balanced by
    atom count (molvoxel_amd/sharding.py); returns this rank's molecules."""
    from molvoxel_amd import sharding
    from molvoxel_amd import workloads as W

    wl = W.cfg4(batch=total)
    bounds = sharding.balanced_shard_bounds([c.shape[0] for c in wl.coords], world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    return wl, lo, hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps; the first ~15 launches after an idle period run up to 18 %% slower (clock ramp)")
    ap.add_argument("--workload", choices=("cfg2", "cfg4"), default="cfg2",
                    help="cfg2 = the headline metric (default); cfg4 = 1024 ligands x world size, sharded by atom count")
    ap.add_argument("--batch", type=int, default=256, help="cfg-2 molecules per GPU per step (8.6 GB of grids at 256)")
    ap.add_argument("--ligands-per-gpu", type=int, default=128, help="cfg-4: the job holds this many ligands per rank")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: mvx_set_overlap - the pre-pass of step k+1 runs under the voxelize launch of step k (the inputs "
                         "are HBM-resident and complete before the loop, which is that mode's contract); 0: serial calls")
    args = ap.parse_args()

    # (RCCL between the ranks of one node needs dmabuf IPC on this driver stack; the pool exports this already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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

    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", device=dev_index,
                                        overlap_prepass=bool(args.overlap))
    if args.workload == "cfg2":
        B = args.batch
        wl, coords, feats = make_batch(B, rank)
        C_ = 32
        kernel_name = "voxelize_kernel<32,gauss>"
        workload = "cfg2: forward_features, gaussian sigma=0.5, scalar radius 1.0, C=32, 64^3, N=4000 atoms/molecule"
        job_molecules = args.gpus * B
    else:
        wl4, lo, hi = make_cfg4_shard(args.ligands_per_gpu * world, rank, world)
        coords, feats = wl4.coords[lo:hi], wl4.channels[lo:hi]
        wl, B, C_ = wl4, hi - lo, 16
        kernel_name = "voxelize_kernel<16,gauss>"
        workload = (f"cfg4: {args.ligands_per_gpu * world} ligands (40-60 atoms), forward_features, gaussian sigma=0.5, "
                    "scalar radius 1.0, C=16, 64^3, sharded by atom count")
        job_molecules = args.ligands_per_gpu * world
    offsets = np.cumsum([0] + [c.shape[0] for c in coords]).astype(np.int64)
    d_coords = vox.asarray(np.concatenate(coords), "coords")
    d_feats = vox.asarray(np.concatenate(feats), "features")
    out = vox.get_empty_grid(C_, batch_size=B)

    def step():
        vox.forward_batch(d_coords, offsets, None, d_feats, 1.0, out_grid=out)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # (event creation takes milliseconds of host time: done before the warm-up, so that the GPU does not sit idle -
    # and drop its clock - between the warm-up and the timed steps)
    vox.set_profiling(True)
    for _ in range(args.warmup):
        step()
    barrier()
    vox.read_kernel_times_ms()  # discard the warm-up launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = vox.read_kernel_times_ms()
    vox.set_profiling(False)
    my_elapsed = elapsed

    rank_ms = [1e3 * my_elapsed / args.steps]
    if dist is not None:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allms = [torch.zeros(1, dtype=torch.float64, device=on) for _ in range(world)]
        dist.all_gather(allms, torch.tensor([1e3 * my_elapsed / args.steps], dtype=torch.float64, device=on))
        rank_ms = [float(x.item()) for x in allms]

    # post-timing spot check of the grids the timed steps left behind (rank 0, outside the timed region)
    spot = None
    if rank == 0:
        picks = sorted({0, B // 2, B - 1})
        spot = parity_spot(out, coords, feats, picks)

    if rank == 0:
        # the library cuts a step's batch into equal chunks of molecules only beyond 65535 (molecule, channel chunk)
        # pairs; per launch = B / launches_per_step molecules x (4*C*D^3 + N*(24 + 4*C + 4))
        lps = max(1, len(kernel_ms) // args.steps)
        step_bytes = sum(4 * C_ * 64**3 + c.shape[0] * (24 + 4 * C_ + 4) for c in coords)
        alg_bytes = step_bytes // lps
        k = np.sort(np.asarray(kernel_ms, dtype=np.float64)) if kernel_ms else np.array([float("nan")])
        k_ms = float(np.mean(k))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        ms_per_step = 1e3 * elapsed / args.steps
        traffic = load_pmc_traffic(B // lps) if args.workload == "cfg2" else None
        res = {
            "metric": "molecules/sec + achieved HBM GB/s, forward_features C=32 64^3 N=4000" if args.workload == "cfg2"
                      else "molecules/sec, batch of ligands (cfg-4), forward_features C=16 64^3 N~50",
            "value": job_molecules * args.steps / elapsed,
            "unit": "molecules/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "molecules_per_gpu_per_step": B,
                "inputs": "HBM-resident (torch CUDA tensors), outputs left in HBM",
                "prepass_overlap": bool(args.overlap),
                "geometry_dtype": "f64",
                "parallelism": f"{args.gpus} independent ranks, molecules sharded, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_ms_avg": k_ms,
                "kernel_ms_min": float(k[0]),
                "kernel_ms_p50": float(k[len(k) // 2]),
                "kernel_ms_max": float(k[-1]),
                "kernel_ms_list": [round(float(x), 4) for x in kernel_ms],
                "kernel_launches_timed": len(kernel_ms),
                "launches_per_step": lps,
                "molecules_per_launch": B // lps,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic": traffic,
                "traffic_source": None if traffic is None else
                                  "profiles/pmc_latest.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command "
                                  "on the builder's box; replayed here, not measured in this run)",
            },
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            "parity_spot": spot,
        }
        if args.gpus == 1 and args.cpu_seconds > 0 and args.workload == "cfg2":
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
