set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/xbin; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for lib in base new; do
  export LIB=molvoxel_amd/csrc/ab/libmvx_$lib.so
  rm -rf $O/tr_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$lib -- python3 tools/cfg2_batch.py 256 > $O/tr_$lib.log 2>&1
  python3 - "$O/tr_$lib" "$lib" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xbin" in r["Name"] or "prep" in r["Name"] or "voxelize" in r["Name"]:
            print(f'[{sys.argv[2]}] {r["Name"][:60]:60s} calls={r["Calls"]:>5s} avg_ns={float(r["AverageNs"]):10.1f}')
PY
done
