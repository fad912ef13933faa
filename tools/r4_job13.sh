set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 800 python3 tools/route_sweep.py chunks 2>&1 | grep -v amdgpu > $O/route_sweep_chunks.txt; cat $O/route_sweep_chunks.txt
for c in cfg2 cfg3 harness; do python3 tools/single_calls.py $c 400 2>&1 | grep -v amdgpu | tail -2; done > $O/single.txt; cat $O/single.txt
