set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4w; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do for spec in "256 radius=1.0" "64 radius=1.5" "64 radius=2.0"; do python3 tools/cfg2_batch.py $spec 2>/dev/null | tail -1; done; done > $O/radius.txt; cat $O/radius.txt
for c in cfg2 cfg3; do python3 tools/single_calls.py $c 400 2>&1 | grep -v amdgpu | tail -2; done > $O/single.txt; cat $O/single.txt
python3 tools/cfg5_single.py 2>&1 | grep -v amdgpu | tail -3
