set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4f
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4f/tests.log 2>&1 || { tail -40 gpurun_out/r4f/tests.log; exit 1; }
tail -3 gpurun_out/r4f/tests.log
for rep in 1 2; do
python3 tools/narrow_rows.py >> gpurun_out/r4f/rows_two.txt 2>&1
NARROW_ONE=1 python3 tools/narrow_rows.py >> gpurun_out/r4f/rows_one.txt 2>&1
done
echo two; cat gpurun_out/r4f/rows_two.txt; echo one; cat gpurun_out/r4f/rows_one.txt
