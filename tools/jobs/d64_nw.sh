cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in 4 8 16; do for A in 50 4000; do for nw in 0 4; do
  if [ $nw = 0 ]; then CHANNELS=$C ATOMS=$A python3 tools/d_kernel_probe.py 64 2>&1 | grep -v amdgpu; else CHANNELS=$C ATOMS=$A NW=$nw python3 tools/d_kernel_probe.py 64 2>&1 | grep -v amdgpu; fi
done; done; done
for C in 4 16; do for nw in 0 3; do if [ $nw = 0 ]; then CHANNELS=$C ATOMS=1000 python3 tools/d_kernel_probe.py 48 2>&1 | grep -v amdgpu; else CHANNELS=$C ATOMS=1000 NW=$nw python3 tools/d_kernel_probe.py 48 2>&1 | grep -v amdgpu; fi; done; done
