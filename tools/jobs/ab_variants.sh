# same-box A/B of library builds over tools/variants.py rows:  AB_NAMES="a b" ONLY="substring" bash tools/jobs/ab_variants.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for name in $AB_NAMES; do echo "[$name rep$rep]"; LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/variants.py 2>&1 | grep -v amdgpu; done; done
for rep in 1 2; do for name in $AB_NAMES; do echo "[$name rep$rep] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py 64 2>/dev/null | tail -1)"; done; done
