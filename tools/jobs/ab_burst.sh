cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in $AB_NAMES $AB_NAMES; do echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/burst.py 256 2 5 2>/dev/null | tail -1)"; echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py 256 2>/dev/null | tail -1)"; done
