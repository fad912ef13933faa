set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; tail -c 900 $O/bench_driverform.json; echo
python3 tools/narrow_rows.py > $O/rows.txt 2>&1; grep -v amdgpu $O/rows.txt
for c in cfg1 cfg2 cfg3 cfg5 harness; do python3 tools/single_calls.py $c 300 2>&1 | grep -v amdgpu | tail -3; done > $O/single.txt; cat $O/single.txt
