# phase timelines of voxelize_kernel from the diagnostic build (tools/ab_build.sh diag "-DMVX_DIAG")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "256 1.0" "64 1.5" "64 2.0"; do set -- $spec; RADIUS=$2 python3 tools/voxelize_timeline.py $1 2>&1 | grep -v amdgpu; echo; done
python3 tools/voxelize_timeline.py cfg5 molvoxel_amd/csrc/ab/libmvx_diag.so 1 2>&1 | grep -v amdgpu; echo
python3 tools/voxelize_timeline.py cfg5 molvoxel_amd/csrc/ab/libmvx_diag.so 4 2>&1 | grep -v amdgpu
