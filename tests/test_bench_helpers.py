"""bench.py pieces that do not need a GPU: the synthetic job, the cpu_baseline leg and the committed PMC summary."""
import json
import os

import numpy as np

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
