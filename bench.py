#!/usr/bin/env python3
"""Headline benchmark: forward_features, C=32, 64^3, N=4000 (BASELINE.json configs[1]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or bare -
    bench.py then starts that launcher itself as a child process, before it touches torch or the GPU)

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
    # all-core C/OpenMP restatement of the same rule, tile-gather form: (x-plane, y-band) tiles (~4 per thread) visit a
    # per-plane atom list, accumulate in a thread-local buffer and write every output row once - the GPU design on cores
    c_oracle.voxelize(xyz, feat, 1.0, dimension=64, out=out)
    m, t0 = 0, time.perf_counter()
    while True:
        c_oracle.voxelize(xyz, feat, 1.0, dimension=64, out=out)
        m += 1
        el2 = time.perf_counter() - t0
        if el2 >= budget_s / 3 or m >= 2000:
            break
    port["openmp_port"] = dict(value=m / el2, unit="molecules/s", cores=c_oracle.num_threads(),
                               sample=f"{m} calls of oracle/mvx_oracle.c (OpenMP tile-gather: per-plane atom lists, thread-local "
                                      f"tile buffers, every row written once) in {el2:.1f} s")
    return port


def parity_spot(out, coords, feats, picks, radius=1.0, dimension=64, sigma=0.5):
    """Post-timing spot check (outside the timed region): the grids of the molecules `picks` of the batch that was
    just timed against the CPU oracle: membership identical and, on every voxel, BOTH bars - the suite's relative rule
    |d| <= 5e-6 * max(1, |ref|) and the north-star's |d| <= 1e-5 absolute (tests/tolerance.py), i.e. the smaller of the two.
    Returns ("ok" or "FAIL: ...", worst |d| seen)."""
    from oracle import c_oracle

    worst_all = 0.0
    for b in picks:
        ref = c_oracle.voxelize(coords[b], feats[b], radius, dimension=dimension, sigma=sigma)
        got = out[b].cpu().numpy()
        bad = int(np.not_equal(got != 0, ref != 0).sum())
        if bad:
            return f"FAIL: molecule {b}: membership differs in {bad} voxels", None
        d = np.abs(got - ref)
        worst_all = max(worst_all, float(d.max()))
        over = d > np.minimum(5e-6 * np.maximum(1.0, np.abs(ref)), 1e-5)
        if over.any():
            return f"FAIL: molecule {b}: {int(over.sum())} voxels beyond min(5e-6 * max(1, |ref|), 1e-5); max |out - ref| = {float(d.max()):.3g}", worst_all
    return "ok", worst_all


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


def measure_pmc_traffic(batch: int, budget_s: float = 90.0):
    """HBM bytes per voxelize launch, MEASURED for this run: two child processes, `rocprofv3 --pmc WRITE_SIZE` and
    `--pmc FETCH_SIZE` (separate passes: the two counters do not fit one) over `bench.py --traffic-probe` - the same
    launch on the same inputs, a few steps, no output. Collected and corrected as MI355X_MICROARCH.md prescribes
    (counters in KB; FETCH_SIZE x 2 on gfx950). Children, not re-executions: this process keeps its GPU context.
    Returns (bytes, detail) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
            or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "this run is itself being profiled"
    got = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-probe", "--batch", str(batch)]
            try:
                res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=budget_s, text=True)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass exceeded {budget_s:.0f} s"
            if res.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {res.returncode})"
            vals = []
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "voxelize_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            vals.append(float(row["Counter_Value"]))
            if len(vals) < 3:
                return None, f"no {counter} rows for voxelize_kernel"
            vals = vals[len(vals) // 2:]  # the later dispatches (warmed up)
            got[counter] = sum(vals) / len(vals) * 1024.0
    total = got["WRITE_SIZE"] + 2.0 * got["FETCH_SIZE"]
    return total, (f"measured in this run: child rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE passes over bench.py --traffic-probe "
                   f"(same launch, {batch} molecules); WRITE_SIZE {got['WRITE_SIZE']:.4g} B + 2 x FETCH_SIZE {got['FETCH_SIZE']:.4g} B "
                   f"(gfx950 correction) per launch")


def traffic_probe(args):
    """--traffic-probe: the timed launch and nothing else (what measure_pmc_traffic profiles)."""
    import torch

    import molvoxel_amd

    torch.cuda.set_device(0)
    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", device=0)
    _, coords, feats = make_batch(args.batch, 0)
    offsets = np.cumsum([0] + [c.shape[0] for c in coords]).astype(np.int64)
    d_coords, d_feats = vox.asarray(np.concatenate(coords), "coords"), vox.asarray(np.concatenate(feats), "features")
    out = vox.get_empty_grid(32, batch_size=args.batch)
    for _ in range(8):
        vox.forward_batch(d_coords, offsets, None, d_feats, 1.0, out_grid=out)
    torch.cuda.synchronize()


def make_cfg4_shard(total: int, rank: int, world: int):
    """cfg-4 (BASELINE.json configs[3]): `total` ligands (40-60 atoms, C = 16) cut into contiguous shards balanced by
    atom count (molvoxel_amd/sharding.py); returns this rank's molecules."""
    from molvoxel_amd import sharding
    from molvoxel_amd import workloads as W

    wl = W.cfg4(batch=total)
    bounds = sharding.balanced_shard_bounds([c.shape[0] for c in wl.coords], world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    return wl, lo, hi


# Default pre-warm (--prewarm N overrides), disclosed in the JSON line (`prewarm_launches`). On a box that has been idle the kernel settles in two
# stages: the first ~15 launches run up to 18 % slower, and the next ~100 still 2-3 % slower (cold box, 30 + 5 warm-up
# launches: kernel 1.425-1.445 ms; the same command a minute later: 1.390-1.417 ms). 200 launches = 0.3 s per rank.
PREWARM_LAUNCHES = 200


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (this process has
    not touched torch or the GPU yet and never will), relay rank 0's single JSON line, return the child's exit code."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)  # stderr passes straight through
    lines = []
    for ln in proc.stdout:
        if ln.startswith("{"):
            lines.append(ln.rstrip("\n"))
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    for ln in lines:
        print(ln, flush=True)
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, saw {len(lines)}\n")
        return 3
    return rc


class Ranks:
    """The only things the ranks share: a barrier, the MAX of the elapsed time and a gather of per-rank figures.
    gloo is always there (CPU, cannot fail on a GPU quirk); when every rank has its own GPU the barrier and the MAX go
    over RCCL ("nccl"), agreed on over gloo so that a rank whose RCCL setup failed cannot leave the others waiting."""

    def __init__(self, world: int, rank: int, device, own_gpu: bool):
        self.world, self.rank, self.device = world, rank, device
        self.dist, self.nccl, self.backend, self.note = None, None, None, None
        # (MVX_BENCH_COLLECTIVES=1 under a one-rank launcher: set the groups up all the same - the way the suite exercises
        # the RCCL branch on a box with a single GPU)
        if world == 1 and os.environ.get("MVX_BENCH_COLLECTIVES") != "1":
            return
        import datetime

        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.backend = dist, "gloo"
        if own_gpu:
            ok, why = 1, ""
            try:
                grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe, group=grp)
                torch.cuda.synchronize()
                ok = int(probe.item() == world)
            except Exception as exc:  # RCCL unusable here: the gloo group carries the barrier instead
                ok, why = 0, f"{type(exc).__name__}: {exc}"[:200]
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.nccl, self.backend = grp, "nccl"
            else:
                self.note = why or "RCCL set-up failed on another rank"

    def barrier(self):
        if self.dist is None:
            return
        if self.nccl is not None:
            import torch

            t = torch.zeros(1, device=self.device)
            self.dist.all_reduce(t, group=self.nccl)
            torch.cuda.synchronize()
        else:
            self.dist.barrier()

    def max_and_gather(self, x: float):
        """(max over ranks, list of every rank's value, number of ranks that answered)"""
        if self.dist is None:
            return x, [x], 1
        import torch

        if self.nccl is not None:
            t = torch.tensor([x], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.nccl)
            mx = float(t.item())
        else:
            t = torch.tensor([x], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            mx = float(t.item())
        every = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(every, torch.tensor([x], dtype=torch.float64))
        seen = torch.ones(1, dtype=torch.int64)
        self.dist.all_reduce(seen)
        return mx, [float(v.item()) for v in every], int(seen.item())

    def gather_objects(self, obj):
        """every rank's small record (device identity, kernel time), in rank order - over gloo"""
        if self.dist is None:
            return [obj]
        every = [None] * self.world
        self.dist.all_gather_object(every, obj)
        return every

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def rehearse(args, world: int, rank: int) -> None:
    """--rehearse: the launcher plumbing of an N-rank run with NO GPU work (for boxes with fewer GPUs than ranks and for
    the CPU test suite): rendezvous, barrier, MAX over ranks, gather, one JSON line from rank 0. Not a measurement."""
    ranks = Ranks(world, rank, None, own_gpu=False)
    acc = np.zeros(1024)
    for _ in range(args.warmup):
        acc += 1.0
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc += 1.0
    ranks.barrier()
    elapsed, rank_s, seen = ranks.max_and_gather(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL of the launcher plumbing - no GPU work, not a measurement", "value": 0.0,
                          "unit": "molecules/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none", "rehearsal": True, "ranks_seen": seen,
                          "collective_backend": ranks.backend, "config": {"workload": "none (plumbing rehearsal)"},
                          "rank_ms_per_step": {"min": 1e3 * min(rank_s) / max(1, args.steps),
                                               "max": 1e3 * max(rank_s) / max(1, args.steps)}}), flush=True)
    ranks.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps before the timed ones (after the fixed pre-warm)")
    ap.add_argument("--workload", choices=("cfg2", "cfg4"), default="cfg2",
                    help="cfg2 = the headline metric (default); cfg4 = 1024 ligands x world size, sharded by atom count")
    ap.add_argument("--batch", type=int, default=256, help="cfg-2 molecules per GPU per step (8.6 GB of grids at 256)")
    ap.add_argument("--ligands-per-gpu", type=int, default=128, help="cfg-4: the job holds this many ligands per rank")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: mvx_set_overlap - the pre-pass of step k+1 runs under the voxelize launch of step k (the inputs "
                         "are HBM-resident and complete before the loop, which is that mode's contract); 0: serial calls")
    ap.add_argument("--pmc-traffic", choices=("auto", "off"), default="auto",
                    help="auto (N = 1, cfg2): measure roofline.traffic in this run with two child rocprofv3 --pmc passes "
                         "(~30 s); if that is not possible, or off: the committed profiles/pmc_latest.json figure, labelled as replayed")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--prewarm", type=int, default=PREWARM_LAUNCHES,
                    help="untimed launches before the counted warm-up (clock ramp; ~1.5 ms each at the default batch)")
    ap.add_argument("--rehearse", action="store_true",
                    help="plumbing only: launcher, rendezvous, barrier, MAX, one JSON line - no GPU work, not a measurement")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: become the launcher. This happens before torch is imported
    # or the GPU is touched, and the ranks are children - nothing is re-executed.
    if args.gpus > 1 and ("WORLD_SIZE" not in os.environ or "RANK" not in os.environ):
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as `python bench.py "
                         f"--gpus {args.gpus}` (it launches its own ranks) or under torch.distributed.run with "
                         f"--nproc-per-node {args.gpus}\n")
        sys.exit(2)
    if args.rehearse:
        return rehearse(args, world, rank)
    if args.traffic_probe:
        return traffic_probe(args)

    # (RCCL between the ranks of one node needs dmabuf IPC on this driver stack; the pool exports this already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    ndev = torch.cuda.device_count()
    # one GPU per rank is the real layout; with fewer GPUs than ranks (rehearsal on a 1-GPU box) ranks share devices
    # and the barrier runs over gloo, because RCCL refuses two ranks on one device
    dev_index = local_rank if ndev >= world else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    ranks = Ranks(world, rank, torch.device("cuda", dev_index), own_gpu=ndev >= world)

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
        ranks.barrier()
        torch.cuda.synchronize()

    # Pass 1 - what `value` and `ms_per_step` come from: no profiling events anywhere near the launches.
    # A fixed pre-warm precedes the counted warm-up: the first ~15 launches after an idle period run up to 18 % slower
    # (clock ramp), and a caller's small --warmup should not decide whether the timed steps sit on that ramp.
    for _ in range(max(0, args.prewarm)):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed, rank_s, ranks_seen = ranks.max_and_gather(my_elapsed)
    rank_ms = [1e3 * s / args.steps for s in rank_s]

    # Pass 2 - the same K steps again, back to back, with the library's HIP events around the voxelize launch (recorded on
    # the launch stream): the kernel durations the roofline figure is built from. The events cost ~11 us of idle GPU per
    # step, which is why this pass does not feed `value`.
    vox.set_profiling(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    vox.read_kernel_times_ms()  # discard
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    profiled_ms_per_step = 1e3 * (time.perf_counter() - t1) / args.steps
    kernel_ms = vox.read_kernel_times_ms()
    vox.set_profiling(False)

    # who ran where, and how fast its kernel was: a scaling efficiency below par can then be pinned on a rank / a device
    props = torch.cuda.get_device_properties(dev_index)
    bus = getattr(props, "pci_bus_id", None)
    me = {"rank": rank, "device_index": dev_index, "device": props.name,
          "pci": None if bus is None else f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}",
          "uuid": str(getattr(props, "uuid", "")), "ms_per_step": 1e3 * my_elapsed / args.steps,
          "kernel_ms_avg": float(np.mean(kernel_ms)) if kernel_ms else None}
    per_rank = ranks.gather_objects(me)
    # ranks that share a device do not measure an N-GPU job: the line says so and carries no `value`
    distinct = len({(r["uuid"], r["pci"], r["device_index"]) for r in per_rank})
    devices_shared = distinct < world

    # post-timing spot check of the grids the timed steps left behind (rank 0, outside the timed region)
    spot, spot_worst = None, None
    if rank == 0:
        picks = sorted({0, B // 2, B - 1})
        spot, spot_worst = parity_spot(out, coords, feats, picks)

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
        traffic, traffic_source = None, None
        if args.workload == "cfg2":
            why = "--pmc-traffic off"
            if args.pmc_traffic == "auto" and args.gpus == 1 and lps == 1:
                traffic, why = measure_pmc_traffic(B)
                traffic_source = why if traffic is not None else None
            if traffic is None:
                traffic = load_pmc_traffic(B // lps)
                if traffic is not None:
                    traffic_source = ("profiles/pmc_latest.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command on the "
                                      f"builder's box; replayed here, not measured in this run: {why})")
        res = {
            "metric": "molecules/sec + achieved HBM GB/s, forward_features C=32 64^3 N=4000" if args.workload == "cfg2"
                      else "molecules/sec, batch of ligands (cfg-4), forward_features C=16 64^3 N~50",
            "value": None if devices_shared else job_molecules * args.steps / elapsed,
            "unit": "molecules/s",
            "n_gpus": args.gpus,
            "devices_shared": devices_shared,
            "distinct_devices": distinct,
            "ranks_seen": ranks_seen,
            "collective_backend": ranks.backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_launches": max(0, args.prewarm),
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
                "parallelism": f"{args.gpus} independent ranks, molecules sharded, no collective on the data path",
                "gpus_visible_per_rank": ndev,
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
                "kernel_ms_source": f"pass 2: the same {args.steps} steps repeated right after the timed ones with HIP events "
                                    "around the voxelize launch on its stream (mvx_set_profiling); `value` is pass 1, no events",
                "profiled_pass_ms_per_step": profiled_ms_per_step,
                "kernel_launches_timed": len(kernel_ms),
                "launches_per_step": lps,
                "molecules_per_launch": B // lps,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic": traffic,
                "traffic_source": traffic_source,
            },
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            "per_rank": per_rank,
            "parity_spot": spot,
            "parity_spot_max_abs": spot_worst,
        }
        if devices_shared:  # (a rehearsal of the N-rank code path on fewer GPUs: the aggregate is NOT an N-GPU figure)
            res["shared_device_molecules_per_s"] = job_molecules * args.steps / elapsed
            res["note"] = (f"{world} ranks on {distinct} device(s): ranks share GPUs, so `value` is withheld; "
                           "shared_device_molecules_per_s is what the shared device(s) delivered")
        if ranks.note:
            res["collective_note"] = ranks.note
        if args.gpus == 1 and args.cpu_seconds > 0 and args.workload == "cfg2":
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
