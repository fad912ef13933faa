set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4c
tools/micro/permlane_swap.bin > gpurun_out/r4c/swap.txt 2>&1 || true; cat gpurun_out/r4c/swap.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4c/tests.log 2>&1 || { tail -30 gpurun_out/r4c/tests.log; exit 1; }
tail -3 gpurun_out/r4c/tests.log
python3 tools/narrow_rows.py > gpurun_out/r4c/rows.txt 2>&1; cat gpurun_out/r4c/rows.txt
bash tools/pmc_cmd.sh r4c_single python3 tools/narrow_rows.py single > gpurun_out/r4c/pmc_single.txt 2>&1; tail -8 gpurun_out/r4c/pmc_single.txt
