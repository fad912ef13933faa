cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_long; mkdir -p $O
timeout -k 10 400 python3 tools/soak.py 800000 14000 > $O/single.txt 2>&1; tail -2 $O/single.txt
timeout -k 10 150 python3 tools/soak.py batches 800000 1000 > $O/batches.txt 2>&1; tail -2 $O/batches.txt
timeout -k 10 200 python3 tools/soak.py routes 800000 200000 > $O/routes.txt 2>&1; tail -2 $O/routes.txt
SOAK_DIMS=4,5,6,7,8,9,11,12,13,17,20,21,28,31,33,36,44,47,49,50,52,56,60,63 timeout -k 10 250 python3 tools/soak.py 900000 7000 > $O/dims.txt 2>&1; tail -2 $O/dims.txt
