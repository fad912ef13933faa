# per-kernel averages (rocprofv3 --kernel-trace --stats) of the narrow rows: what the pre-pass costs next to the voxelize launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prepass; mkdir -p $O
for row in single types8 cfg3x256 cfg1x256; do
  rm -rf $O/tr_$row
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$row -- python3 tools/narrow_rows.py $row > $O/tr_$row.log 2>&1
  grep -v amdgpu $O/tr_$row.log | tail -1
  python3 - "$O/tr_$row" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("xbin", "prep", "voxelize", "chan_aux")):
            print(f'    {r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
done
