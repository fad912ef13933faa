cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CHANNELS=16 ATOMS=4000
python3 tools/d_kernel_probe.py 64 96 128 2>&1 | grep -v amdgpu
NW=12 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
NW=4 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
NW=6 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
export ATOMS=50
python3 tools/d_kernel_probe.py 64 96 128 2>&1 | grep -v amdgpu
NW=12 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
