# GPU suite, then kernel / call times on odd and even grid sizes for narrow and wide launches (tools/d_kernel_probe.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/oddprobe; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for C in 1 4 8 16; do CHANNELS=$C python3 tools/d_kernel_probe.py 48 49 50 63 64 2>&1 | grep -v amdgpu; done
