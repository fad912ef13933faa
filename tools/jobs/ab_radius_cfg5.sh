set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
rm -f $O/ab.txt; bash tools/ab_radius.sh $O/ab.txt base new; cat $O/ab.txt
for lib in base new base new; do echo "[$lib] $(python3 tools/cfg5_single.py molvoxel_amd/csrc/ab/libmvx_$lib.so 2>/dev/null | tail -1)"; done
for lib in base new base new; do echo "[$lib] $(python3 tools/cfg5_single.py molvoxel_amd/csrc/ab/libmvx_$lib.so 0 0 4 2>/dev/null | tail -1)"; done
