# one test selection on the GPU box:  TESTSEL="tests/test_hip_parity.py -k name" bash tools/jobs/one_test.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest $TESTSEL -m gpu -x -q 2>&1 | tail -30
