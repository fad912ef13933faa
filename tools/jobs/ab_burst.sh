cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in $AB_NAMES $AB_NAMES; do for B in 128 256; do echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/burst.py $B 2 5 2>/dev/null | tail -1)"; done; done
