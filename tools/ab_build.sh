#!/bin/bash
# Builds A/B variants of libmvx_hip.so:  tools/ab_build.sh <name> "<extra -D flags>"  -> molvoxel_amd/csrc/ab/libmvx_<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../molvoxel_amd/csrc"
mkdir -p ab
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed -Wno-bitwise-instead-of-logical $flags"
/opt/rocm/bin/hipcc $F -c -o ab/k_$name.o mvx_kernels.hip &
/opt/rocm/bin/hipcc $F -c -o ab/c_$name.o mvx_capi.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmvx_$name.so ab/k_$name.o ab/c_$name.o
rm -f ab/k_$name.o ab/c_$name.o
echo built ab/libmvx_$name.so
