cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak; mkdir -p $O
timeout -k 10 280 python3 tools/soak.py 600000 6000 > $O/single.txt 2>&1; tail -2 $O/single.txt
timeout -k 10 200 python3 tools/soak.py batches 600000 300 > $O/batches.txt 2>&1; tail -2 $O/batches.txt
timeout -k 10 330 python3 tools/soak.py routes 600000 30000 > $O/routes.txt 2>&1; tail -2 $O/routes.txt
SOAK_DIMS=4,6,7,8,9,12,13,20,28,33,36,44,49,50,52,56,60,63 timeout -k 10 200 python3 tools/soak.py 700000 2500 > $O/dims.txt 2>&1; tail -2 $O/dims.txt
