#!/bin/bash
# Host-side sanitizer run (the GPU pool offers none): mvx_plan.hip built with UBSan (host pass only), 300 000 random queries.
set -e
cd "$(dirname "$0")/../molvoxel_amd/csrc"
mkdir -p ab
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -fsanitize=undefined -fno-sanitize-recover=undefined -Wno-unused-function -Wno-option-ignored -c -o ab/plan_ubsan.o mvx_plan.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=undefined -Wno-option-ignored -o ab/libplan_ubsan.so ab/plan_ubsan.o
rm -f ab/plan_ubsan.o
cd ../.. && python3 tools/plan_fuzz.py
