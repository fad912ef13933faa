#!/bin/bash
# same-box A/B of library builds over the radius sweep:  tools/ab_radius.sh <out-file> <name> [<name> ...]   (molvoxel_amd/csrc/ab/libmvx_<name>.so)
out=$1; shift
for rep in 1 2; do
  for name in "$@"; do
    for spec in "256 radius=1.0" "64 radius=1.5" "64 radius=2.0"; do
      echo -n "[$name rep$rep] " >> $out
      LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py $spec 2>/dev/null | tail -1 >> $out
    done
  done
done
