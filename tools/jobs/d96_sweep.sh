# D = 96 (rows of 1.5 x 256 B: the last z chunk of 8-wave slabs is half empty): slab plans over channels and densities
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in 4 8 16 32 64; do for A in 50 4000 13500; do for nw in 0 4 12; do
  if [ $nw = 0 ]; then CHANNELS=$C ATOMS=$A python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu; else CHANNELS=$C ATOMS=$A NW=$nw python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu; fi
done; done; done
