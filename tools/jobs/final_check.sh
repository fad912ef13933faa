# the driver's round-end sequence on one box: GPU suite, smoke(), default bench line; then the operator variants
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
python3 tools/variants.py 2>&1 | grep -v amdgpu > $O/variants.txt; grep "D=96\|D=48\|cfg-2 (" $O/variants.txt
