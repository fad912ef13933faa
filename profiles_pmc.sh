#!/bin/bash
# usage: profiles_pmc.sh <outdir> -- collects PMC passes for bench.py (short run)
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/pmc_summary.txt").read())
PY
