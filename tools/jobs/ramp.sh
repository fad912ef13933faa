cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in cur nopace cur nopace; do echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/burst_ramp.py 256 16 0.5 2>/dev/null | tail -1)"; done
for name in cur nopace; do echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/burst_ramp.py 256 12 0.05 2>/dev/null | tail -1)"; done
