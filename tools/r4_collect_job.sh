cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_driverform.json 2> $O/bench_driverform.err; tail -c 600 $O/bench_driverform.json; echo
STEPS=20 WARMUP=5 bash profiles/collect.sh r04 > $O/collect.log 2>&1; tail -3 $O/collect.log
python3 bench_configs.py --harness > $O/bench_configs.jsonl 2> $O/bench_configs.err; cat $O/bench_configs.jsonl
python3 bench_configs.py --pacing > $O/pacing.jsonl 2>> $O/bench_configs.err; cat $O/pacing.jsonl
python3 tools/variants.py > $O/variants.txt 2>&1; grep -v amdgpu $O/variants.txt
for k in 1 4 8 16 32; do python3 tools/chanwise_probe.py $k; done 2>&1 | grep -v amdgpu > $O/chanwise.txt; cat $O/chanwise.txt
for spec in "256 radius=1.0" "64 radius=1.5" "64 radius=2.0"; do python3 tools/cfg2_batch.py $spec 2>/dev/null | tail -1; done > $O/radius.txt; cat $O/radius.txt
bash tools/pmc_cmd.sh r4h_single python3 tools/narrow_rows.py single > $O/pmc_single.txt 2>&1; tail -9 $O/pmc_single.txt
bash tools/pmc_cmd.sh r4h_cfg3 python3 tools/narrow_rows.py cfg3x256 > $O/pmc_cfg3.txt 2>&1; tail -9 $O/pmc_cfg3.txt
python3 tools/narrow_rows.py > $O/rows.txt 2>&1; grep -v amdgpu $O/rows.txt
python3 bench.py --workload cfg4 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_cfg4.json 2>/dev/null; tail -c 400 $O/bench_cfg4.json; echo
