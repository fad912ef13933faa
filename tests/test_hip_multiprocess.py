"""The N > 1 layout on real hardware: two processes launched by torch.distributed.run (one rank per GPU; on the
1-GPU test box both ranks share the device and rendezvous over gloo), each running the HIP path on its shard.

What a scaling run on an 8-GPU node relies on and a 1-GPU box can still check: the launcher contract
(RANK / LOCAL_RANK / WORLD_SIZE, 127.0.0.1 rendezvous), the atom-count-balanced partition tiling the job exactly
once, every rank's grids equal to the oracle's, two processes sharing one libmvx_hip.so build without interfering,
and bench.py's own N = 2 code path (barrier, MAX over ranks, one JSON line from rank 0, parity spot check).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script_args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)


def test_two_ranks_voxelize_their_shards_of_cfg4(tmp_path):
    from molvoxel_amd import workloads as W

    total = 24
    out = tmp_path / "shards.json"
    res = _torchrun(2, [os.path.join(ROOT, "tests", "mp_shard_worker.py"), str(total), str(out)])
    assert res.returncode == 0, res.stdout[-3000:]
    rep = json.load(open(out))
    assert rep["world"] == 2 and rep["max_elapsed"] > 0
    covered, sums = [], []
    for r in sorted(rep["ranks"], key=lambda r: r["rank"]):
        covered += list(range(r["lo"], r["hi"]))
        sums += r["sums"]
        assert r["worst"] <= 5e-6
    assert covered == list(range(total)), "shards must tile the batch exactly once"
    wl = W.cfg4(batch=total)
    atoms = [r["atoms"] for r in rep["ranks"]]
    assert sum(atoms) == sum(c.shape[0] for c in wl.coords)
    assert abs(atoms[0] - atoms[1]) <= 60, "atom-count-balanced shards differ by at most one ligand"
    # the same molecules in one process give the same sums (bitwise: same kernels, same order)
    import molvoxel_amd

    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip")
    offsets = np.cumsum([0] + [c.shape[0] for c in wl.coords]).astype(np.int64)
    grid = vox.forward_batch(vox.asarray(np.concatenate(wl.coords), "coords"), offsets, None,
                             vox.asarray(np.concatenate(wl.channels), "features"), 1.0)
    single = [float(grid[b].cpu().numpy().sum(dtype=np.float64)) for b in range(total)]
    assert single == sums


@pytest.mark.parametrize("workload,extra", [("cfg2", ["--batch", "8"]), ("cfg4", ["--ligands-per-gpu", "16"])])
def test_bench_py_two_rank_code_path(workload, extra):
    res = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload,
                        "--cpu-seconds", "0"] + extra)
    assert res.returncode == 0, res.stdout[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == "weak" and rec["parity_spot"] == "ok"
    _check_shared_device_line(rec, 2)
    assert rec["rank_ms_per_step"]["max"] >= rec["rank_ms_per_step"]["min"] > 0
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1.2


def _check_shared_device_line(rec, n):
    """On this 1-GPU box the ranks share the card: the line must say so and must NOT carry a `value` (an aggregate over
    ranks that share a GPU is not an N-GPU measurement); every rank reports where it ran and its kernel time."""
    import torch

    shared = torch.cuda.device_count() < n
    assert rec["devices_shared"] == shared and rec["distinct_devices"] == (1 if shared else n)
    if shared:
        assert rec["value"] is None and rec["shared_device_molecules_per_s"] > 0 and "share" in rec["note"]
    else:
        assert rec["value"] > 0
    assert [r["rank"] for r in rec["per_rank"]] == list(range(n))
    for r in rec["per_rank"]:
        assert r["kernel_ms_avg"] > 0 and r["ms_per_step"] > 0 and r["device"] and r["device_index"] >= 0
    assert rec["parity_spot_max_abs"] is not None and rec["parity_spot_max_abs"] <= 1e-5


def _bare_bench(args, timeout=900):
    """bench.py with NO launcher and no rank variables in the environment: it must start its own ranks."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, text=True)


@pytest.mark.parametrize("n,extra", [(2, ["--batch", "8"]), (4, ["--batch", "4"]), (4, ["--workload", "cfg4", "--ligands-per-gpu", "8"])])
def test_bare_bench_py_starts_its_own_ranks_on_the_hip_path(n, extra):
    """The driver's N-GPU command without the launcher (`python3 bench.py --gpus N ...`): one JSON line, every rank
    counted, parity spot check green. On this 1-GPU box the ranks share the device (at most 4 here: the pool allows six
    processes on a card) and rendezvous over gloo; the 8-rank form is rehearsed without GPU work in test_bench_helpers.py."""
    res = _bare_bench(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--prewarm", "5", "--cpu-seconds", "0"] + extra)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = res.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["ranks_seen"] == n and rec["parity_spot"] == "ok" and rec["scaling"] == "weak"
    assert rec["collective_backend"] in ("gloo", "nccl") and rec["prewarm_launches"] >= 1
    _check_shared_device_line(rec, n)
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1.2


def test_bench_py_rccl_branch_with_one_rank():
    """The barrier / MAX over RCCL - what an 8-GPU run uses when every rank has its own GPU - cannot be reached with two
    ranks on one card (RCCL refuses two ranks per device). One rank under the launcher with the collectives forced on
    runs exactly that code: gloo group, RCCL sub-group, probe all-reduce, agreement, barrier, MAX, gather."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MVX_BENCH_COLLECTIVES"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--cpu-seconds", "0", "--pmc-traffic", "off"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["collective_backend"] == "nccl" and rec["ranks_seen"] == 1 and rec["parity_spot"] == "ok", rec.get("collective_note")


def test_bench_py_measures_its_hbm_traffic_in_the_run():
    """roofline.traffic of the N = 1 line: two child `rocprofv3 --pmc` passes (WRITE_SIZE, FETCH_SIZE) over the same
    launch, in this run - not a figure replayed from profiles/."""
    res = _bare_bench(["--steps", "2", "--warmup", "1", "--batch", "8", "--cpu-seconds", "0"])
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    rec = json.loads(res.stdout.splitlines()[-1])
    rf = rec["roofline"]
    assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    assert 1.0 <= rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.25
