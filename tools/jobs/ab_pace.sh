# same-box A/B of library builds over launch sizes (the pacing thresholds): cfg-2 x 16 / 32 / 64, cfg-5 x 1 / 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for name in $AB_NAMES; do
    for B in 8 16 32 64; do echo "[$name rep$rep] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py $B 2>/dev/null | tail -1)"; done
    echo "[$name rep$rep] $(python3 tools/cfg5_single.py molvoxel_amd/csrc/ab/libmvx_$name.so 2>/dev/null | tail -1)"
    echo "[$name rep$rep] $(python3 tools/cfg5_single.py molvoxel_amd/csrc/ab/libmvx_$name.so 0 0 4 2>/dev/null | tail -1)"
  done
done
