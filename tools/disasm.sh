#!/bin/bash
# tools/disasm.sh <mangled-name-substring> [object] [lines] -> /tmp/kernel.s (the kernel's ISA), prints where scratch / barriers / loads / stores sit
obj=${2:-/root/repo/molvoxel_amd/csrc/mvx_slab.o}
L=/opt/rocm/lib/llvm/bin
cd /tmp && $L/llvm-objcopy --dump-section=.hip_fatbin=/tmp/fat.bin $obj /tmp/ign && $L/clang-offload-bundler --unbundle --type=o --input=/tmp/fat.bin --output=/tmp/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 && $L/llvm-objdump -d --no-show-raw-insn /tmp/k.co > /tmp/k.s
S=$(grep -n "^[0-9a-f]* <.*$1" /tmp/k.s | head -1 | cut -d: -f1)
awk -v s=$S 'NR>=s' /tmp/k.s | awk 'NR>1 && /^[0-9a-f]+ <_Z/{exit} {print}' | sed 's#//.*##' > /tmp/kernel.s
wc -l /tmp/kernel.s
grep -n "scratch_\|s_barrier\|v_exp_f32\|global_store_dwordx4\|global_load\|s_endpgm\|s_sleep" /tmp/kernel.s | awk '{print $1, $2, $3, $4}' | head -${3:-80}
