# GPU suite, then narrow launches on long rows under the shipped plan (tools/d_kernel_probe.py) and a soak over the long-row grid sizes
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bigd; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for C in 1 4; do CHANNELS=$C python3 tools/d_kernel_probe.py 72 88 100 104 120 2>&1 | grep -v amdgpu; done
SOAK_DIMS=68,72,76,84,88,92,100,104,108,112,116,120,124 timeout -k 10 400 python3 tools/soak.py 1400000 2500 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
