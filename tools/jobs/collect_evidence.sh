cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evidence; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; tail -c 600 $O/bench_driverform.json; echo
STEPS=20 WARMUP=5 bash profiles/collect.sh r04 > $O/collect.log 2>&1; tail -3 $O/collect.log
python3 bench_configs.py --harness > $O/bench_configs.jsonl 2> $O/bench_configs.err; cat $O/bench_configs.jsonl
python3 bench_configs.py --pacing > $O/pacing.jsonl 2>> $O/bench_configs.err; cat $O/pacing.jsonl
python3 tools/variants.py > $O/variants.txt 2>&1; grep -v amdgpu $O/variants.txt
python3 tools/narrow_rows.py > $O/rows.txt 2>&1; grep -v amdgpu $O/rows.txt
python3 tools/route_sweep.py all 2>&1 | grep -v amdgpu > $O/route_sweep.txt; tail -5 $O/route_sweep.txt
# rocprofv3 kernel trace of per-molecule calls (the pair kernel's average duration next to the HIP-event figure)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/single_trace -- python3 tools/single_calls.py cfg2 400 > $O/single_trace.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/evidence/single_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_ns={float(r["AverageNs"]):10.1f} pct={r["Percentage"]}')
PY
