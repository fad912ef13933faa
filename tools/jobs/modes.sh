# the headline kernel's two modes (0.81 / 0.83 of peak between processes on one box): distribution, and the allocator's part in it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6; do echo "[default] $(python3 tools/cfg2_batch.py 256 2>/dev/null | tail -1)"; done
for i in 1 2 3 4; do echo "[expandable] $(PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python3 tools/cfg2_batch.py 256 2>/dev/null | tail -1)"; done
for i in 1 2 3 4; do echo "[no caching] $(PYTORCH_NO_HIP_MEMORY_CACHING=1 python3 tools/cfg2_batch.py 256 2>/dev/null | tail -1)"; done
