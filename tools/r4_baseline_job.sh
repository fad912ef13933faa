set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4base
python3 tools/narrow_rows.py > gpurun_out/r4base/rows.txt 2>&1
for row in single types8 feat5 feat16 cfg3x256 cfg1x256; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4base/kt_$row -- python3 tools/narrow_rows.py $row > gpurun_out/r4base/kt_$row.log 2>&1
  echo "== $row" >> gpurun_out/r4base/kstats.txt
  python3 tools/kstats.py gpurun_out/r4base/kt_$row >> gpurun_out/r4base/kstats.txt
done
bash tools/pmc_cmd.sh r4base_single python3 tools/narrow_rows.py single > gpurun_out/r4base/pmc_single.txt 2>&1
bash tools/pmc_cmd.sh r4base_cfg3 python3 tools/narrow_rows.py cfg3x256 > gpurun_out/r4base/pmc_cfg3.txt 2>&1
cat gpurun_out/r4base/rows.txt gpurun_out/r4base/kstats.txt
