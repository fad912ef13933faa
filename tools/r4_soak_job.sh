cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4soak; mkdir -p $O
timeout -k 10 280 python3 tools/soak.py 400000 6000 > $O/single.txt 2>&1; tail -2 $O/single.txt
timeout -k 10 200 python3 tools/soak.py batches 400000 300 > $O/batches.txt 2>&1; tail -2 $O/batches.txt
timeout -k 10 280 python3 tools/soak.py routes 400000 20000 > $O/routes.txt 2>&1; tail -2 $O/routes.txt
SOAK_DIMS=6,7,9,13,20,28,36,44,49,52,56,60,63,65,66,72,88,100,112,128 timeout -k 10 200 python3 tools/soak.py 500000 2500 > $O/dims.txt 2>&1; tail -2 $O/dims.txt
