"""bench.py pieces that do not need a GPU: the synthetic job, the cpu_baseline leg and the committed PMC summary."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_job_is_deterministic_and_sharded_by_rank():
    """Molecule j of the job is the same whichever rank owns it (weak scaling: rank r owns [r*B, (r+1)*B))."""
    wl, c0, f0 = bench.make_batch(2, rank=0)
    _, c1, f1 = bench.make_batch(1, rank=1)
    assert wl.dimension == 64 and len(c0) == 2 and c0[0].shape == (4000, 3) and f0[0].shape == (4000, 32)
    assert np.array_equal(c0[1], c1[0]) and np.array_equal(f0[1], f1[0]) and not np.array_equal(c0[0], c0[1])
    # B = 4*C*D^3 + N*(3*8 + 4*C + 4)
    assert wl.algorithmic_bytes(0) == 4 * 32 * 64**3 + 4000 * (24 + 4 * 32 + 4)


def test_cpu_baseline_leg_reports_the_contract_fields():
    r = bench.cpu_baseline(0.3)
    assert r["kind"] == "port" and r["unit"] == "molecules/s" and r["value"] > 0 and r["cores"] >= 1
    assert "numpy_port" in r["sample"] and r["openmp_port"]["value"] > 0


def test_committed_pmc_summary_matches_the_default_workload():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    per_launch = d["molecules_per_launch"]
    assert bench.load_pmc_traffic(per_launch) == d["hbm_bytes_per_launch"]
    assert bench.load_pmc_traffic(per_launch + 1) is None
    alg = per_launch * (4 * 32 * 64**3 + 4000 * (24 + 4 * 32 + 4))
    assert 1.0 <= d["hbm_bytes_per_launch"] / alg < 1.1  # traffic close to the algorithmic bytes: no wasted re-reads
    for name in ("r01_bench_line.json", "r02_bench_line.json"):
        line = json.load(open(os.path.join(ROOT, "profiles", name)))
        _check_line(line)
    assert line["parity_spot"] == "ok" and 0.5 < line["roofline"]["step_frac"] < line["roofline"]["frac"] < 1.0


def _check_line(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}


def _bench(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_bare_bench_launches_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher around it starts torch.distributed.run as a child, relays rank 0's one
    JSON line on stdout (nothing else there) and exits with the child's code. --rehearse: plumbing only, no GPU work -
    the N = 8 form of the driver's scaling run, rehearsed where there is no 8-GPU node."""
    res = _bench(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--rehearse"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = res.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["ranks_seen"] == n and rec["rehearsal"] is True and rec["collective_backend"] == "gloo"
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["rank_ms_per_step"]["max"] >= rec["rank_ms_per_step"]["min"] > 0


def test_bench_refuses_a_launcher_of_the_wrong_size():
    res = _bench(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert res.returncode == 2 and "WORLD_SIZE=3" in res.stderr and res.stdout == ""


def test_bare_bench_relays_the_exit_code_of_failed_ranks():
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without a GPU: the ranks must fail")
    res = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--batch", "1"])
    assert res.returncode != 0 and res.stdout == ""


def test_live_traffic_measurement_degrades_to_the_committed_figure():
    """Without a GPU the child rocprofv3 passes cannot run: the helper says why instead of raising, and bench.py then
    labels the committed figure as replayed."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    traffic, why = bench.measure_pmc_traffic(1, budget_s=120.0)
    assert traffic is None and isinstance(why, str) and why
