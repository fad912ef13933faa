# same-box A/B of library builds on ligand batches (cfg-4 x 128: mostly empty slabs) and, as a guard, the headline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for name in $AB_NAMES; do
  echo "[$name rep$rep] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/run_cfg.py cfg4 128 60 2>/dev/null | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('cfg4 x128 kernel %.4f ms (%.0f GB/s) call %.4f ms' % (d['kernel_ms'], d['GBps'], d['ms_per_call']))")"
done; done
for name in $AB_NAMES; do echo "[$name] $(LIB=molvoxel_amd/csrc/ab/libmvx_$name.so python3 tools/cfg2_batch.py 256 2>/dev/null | tail -1)"; done
