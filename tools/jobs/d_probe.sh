cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/d_kernel_probe.py 64 72 80 96 112 128 2>&1 | grep -v amdgpu
NW=12 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
NW=6 python3 tools/d_kernel_probe.py 96 2>&1 | grep -v amdgpu
NW=4 python3 tools/d_kernel_probe.py 96 128 2>&1 | grep -v amdgpu
NW=16 python3 tools/d_kernel_probe.py 128 2>&1 | grep -v amdgpu
