set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/single; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do for c in cfg1 cfg2 cfg3 harness; do python3 tools/single_calls.py $c 400 2>&1 | grep -v amdgpu | tail -2; done; done > $O/single.txt; cat $O/single.txt
