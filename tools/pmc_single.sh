#!/bin/bash
# SQ counters of per-molecule calls:  tools/pmc_single.sh <cfg> <tag>
set -u
cfg=${1:-cfg2}; tag=${2:-x}
out=gpurun_out/pmc_single_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 tools/single_calls.py $cfg 100 > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "mvx" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY
