# rows of an odd number of sub-tiles below 64 voxels (D = 24, 40, 56) in one- / four-channel launches: whole rows (plan) against even chunks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in 1 4 8; do
  for spec in "24 2" "40 2 4" "56 4 6"; do
    set -- $spec; D=$1; shift
    CHANNELS=$C python3 tools/d_kernel_probe.py $D 2>&1 | grep -v amdgpu
    for nw in "$@"; do CHANNELS=$C NW=$nw python3 tools/d_kernel_probe.py $D 2>&1 | grep -v amdgpu; done
  done
done
