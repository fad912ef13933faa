#!/bin/bash
# Builds A/B variants of libmvx_hip.so:  tools/ab_build.sh <name> "<extra -D flags>"  -> molvoxel_amd/csrc/ab/libmvx_<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../molvoxel_amd/csrc"
mkdir -p ab
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed -Wno-bitwise-instead-of-logical $flags"
objs=""
for tu in capi plan prep slab direct pair f64 splat; do
  [ -f mvx_$tu.hip ] || continue
  /opt/rocm/bin/hipcc $F -c -o ab/${tu}_$name.o mvx_$tu.hip &
  objs="$objs ab/${tu}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmvx_$name.so $objs
rm -f $objs
echo built ab/libmvx_$name.so
