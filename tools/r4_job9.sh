set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for c in cfg1 cfg2 cfg3 cfg5 harness; do python3 tools/single_calls.py $c 300 2>&1 | grep -v amdgpu | tail -3; done > $O/single.txt; cat $O/single.txt
for c in cfg2 harness cfg3 cfg1; do timeout -k 10 120 python3 tools/direct_timeline.py $c 2>&1 | grep -v amdgpu; done > $O/direct_timeline.txt 2>&1; cat $O/direct_timeline.txt
