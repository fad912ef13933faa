# same-box A/B/C of library builds (molvoxel_amd/csrc/ab/libmvx_<name>.so): headline + radius rows + cfg-5
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab3; mkdir -p $O
rm -f $O/ab.txt; bash tools/ab_radius.sh $O/ab.txt $AB_NAMES; cat $O/ab.txt
for lib in $AB_NAMES $AB_NAMES; do echo "[$lib] $(python3 tools/cfg5_single.py molvoxel_amd/csrc/ab/libmvx_$lib.so 2>/dev/null | tail -1)"; done
