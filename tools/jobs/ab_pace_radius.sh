cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for name in $AB_NAMES; do for spec in "128 radius=1.5" "128 radius=2.0" "256 radius=1.25"; do echo "[$name rep$rep] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py $spec 2>/dev/null | tail -1)"; done; done; done
