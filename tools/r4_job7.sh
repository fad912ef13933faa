set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4g
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4g/tests.log 2>&1 || { tail -40 gpurun_out/r4g/tests.log; exit 1; }
tail -3 gpurun_out/r4g/tests.log
for rep in 1 2; do
python3 tools/narrow_rows.py single types8 feat5 cfg3x256 >> gpurun_out/r4g/rows_rule.txt 2>&1
NARROW_SUB=2 python3 tools/narrow_rows.py single types8 feat5 cfg3x256 >> gpurun_out/r4g/rows_2.txt 2>&1
done
echo rule; cat gpurun_out/r4g/rows_rule.txt; echo two; cat gpurun_out/r4g/rows_2.txt
bash tools/pmc_cmd.sh r4g_single python3 tools/narrow_rows.py single > gpurun_out/r4g/pmc_single.txt 2>&1; tail -9 gpurun_out/r4g/pmc_single.txt
bash tools/pmc_cmd.sh r4g_cfg3 python3 tools/narrow_rows.py cfg3x256 > gpurun_out/r4g/pmc_cfg3.txt 2>&1; tail -9 gpurun_out/r4g/pmc_cfg3.txt
