# bench.py with the cross-call pre-pass overlap (opt-in, never the driver's `value`), then the operator variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/overlap; mkdir -p $O
for ov in 0 1 0 1; do python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pmc-traffic off --overlap $ov 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap', d.get('prepass_overlap'), 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'kernel frac', round(d['roofline']['frac'],3), 'step_frac', round(d['roofline']['step_frac'],3))"; done
python3 tools/variants.py 2>&1 | grep -v amdgpu > $O/variants.txt; grep "D=48" $O/variants.txt
