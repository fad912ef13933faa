set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4e
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4e/tests.log 2>&1 || { tail -40 gpurun_out/r4e/tests.log; exit 1; }
tail -3 gpurun_out/r4e/tests.log
for cfg in cfg1 cfg2 cfg3 cfg5 harness; do
  echo "== $cfg default" >> gpurun_out/r4e/single.txt; python3 tools/single_calls.py $cfg 300 2>/dev/null >> gpurun_out/r4e/single.txt
  echo "== $cfg direct=0 fused=0" >> gpurun_out/r4e/single.txt; DIRECT=0 FUSED=0 python3 tools/single_calls.py $cfg 300 2>/dev/null >> gpurun_out/r4e/single.txt
  echo "== $cfg direct=0 fused=1" >> gpurun_out/r4e/single.txt; DIRECT=0 FUSED=1 python3 tools/single_calls.py $cfg 300 2>/dev/null >> gpurun_out/r4e/single.txt
done
cat gpurun_out/r4e/single.txt
