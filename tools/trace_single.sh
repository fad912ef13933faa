#!/bin/bash
# rocprofv3 --kernel-trace --stats of per-molecule forward() calls, one run per configuration.
#   tools/trace_single.sh <tag>   -> gpurun_out/single_<tag>/<cfg>.txt
set -u
tag=${1:-x}
out=gpurun_out/single_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for cfg in cfg1 cfg2 cfg3 cfg5 harness; do
  python3 tools/single_calls.py $cfg 300 > "$out/$cfg.host.txt" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$cfg" -- python3 tools/single_calls.py $cfg 300 > "$out/$cfg.log" 2>&1 || echo "trace $cfg failed"
  python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/{cfg}/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
with open(f"{out}/{cfg}.txt", "w") as fh:
    fh.write(open(f"{out}/{cfg}.host.txt").read())
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        fh.write(f'{r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_ns={float(r["AverageNs"]):10.1f} pct={r["Percentage"]}\n')
print(open(f"{out}/{cfg}.txt").read())
PY
done
