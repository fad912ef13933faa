set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4d
for rep in 1 2; do
python3 tools/narrow_rows.py >> gpurun_out/r4d/rows_main.txt 2>&1
LIB=molvoxel_amd/csrc/ab/libmvx_nocull.so python3 tools/narrow_rows.py >> gpurun_out/r4d/rows_nocull.txt 2>&1
done
echo main; cat gpurun_out/r4d/rows_main.txt; echo nocull; cat gpurun_out/r4d/rows_nocull.txt
