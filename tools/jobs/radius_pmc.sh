# counters of voxelize_kernel at radius 1.5 / 2.0 A (cfg-2 geometry x 64), incl. scalar-unit and matrix-pipe activity
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/radpmc; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_[A-Z_0-9]*SCA[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_[A-Z_0-9]*VALU[A-Z_0-9]*" $O/counters.txt | sort -u > $O/names.txt || true
cat $O/names.txt
for R in 2.0 1.5; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
             "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/r$R/p$i -- python3 tools/cfg2_batch.py 64 radius=$R > $O/r${R}_p$i.log 2>&1 || echo "pass $i R=$R failed: $(tail -3 $O/r${R}_p$i.log)"
  done
done
python3 - $O <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for R in ("2.0", "1.5"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/r{R}/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if "voxelize_kernel" not in k: continue
        print("radius", R, k)
        for c in sorted(d):
            v = d[c][len(d[c]) // 2:]
            print(f"   {c:32s} n={len(d[c]):4d} mean={sum(v) / len(v):.6g}")
PY
