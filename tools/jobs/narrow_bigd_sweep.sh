# rows of 65 ... 128 voxels in narrow launches (C = 1 / 4 / 8: the multi-sub-tile kernel needs an even number of waves per slab):
# whole-row slabs of 9 ... 16 waves (NW plan) against chunks of 4 / 6 / 8 sub-tiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in ${CLIST:-1 4 8}; do
  for D in 72 80 88 104 112 120; do
    CHANNELS=$C python3 tools/d_kernel_probe.py $D 2>&1 | grep -v amdgpu
    for nw in 4 6 8; do CHANNELS=$C NW=$nw python3 tools/d_kernel_probe.py $D 2>&1 | grep -v amdgpu; done
  done
done
