# same-box A/B of library builds over the narrow rows (tools/narrow_rows.py; checksums must agree)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for name in $AB_NAMES; do echo "[$name rep$rep]"; LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/narrow_rows.py 2>&1 | grep -v amdgpu; done; done
