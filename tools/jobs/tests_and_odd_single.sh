set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/odd; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/odd_single.py 2>&1 | grep -v amdgpu > $O/odd_single.txt; cat $O/odd_single.txt
