set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4b/tests.log 2>&1 || { tail -30 gpurun_out/r4b/tests.log; exit 1; }
tail -3 gpurun_out/r4b/tests.log
python3 tools/narrow_rows.py > gpurun_out/r4b/rows.txt 2>&1; cat gpurun_out/r4b/rows.txt
python3 tools/variants.py > gpurun_out/r4b/variants.txt 2>&1; cat gpurun_out/r4b/variants.txt
for k in 1 4 8 16 32; do python3 tools/chanwise_probe.py $k; done > gpurun_out/r4b/chanwise.txt 2>&1; cat gpurun_out/r4b/chanwise.txt
python3 tools/cfg2_batch.py 256 > gpurun_out/r4b/cfg2.txt 2>&1; python3 tools/cfg2_batch.py 64 radius=2.0 >> gpurun_out/r4b/cfg2.txt 2>&1; cat gpurun_out/r4b/cfg2.txt
