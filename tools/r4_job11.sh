set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_pair_kernel.py -m gpu -x -q > $O/tests_pair.log 2>&1 || { tail -60 $O/tests_pair.log; exit 1; }
tail -3 $O/tests_pair.log
python3 tools/host_profile.py cfg3 3000 2>&1 | grep -v amdgpu > $O/host_profile.txt; cat $O/host_profile.txt
