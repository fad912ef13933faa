#!/bin/bash
# SQ / memory counters of voxelize_kernel for any command:  tools/pmc_cmd.sh <tag> python3 <script> [args...]
# (the program itself must follow, not a shell or env wrapper: the profiler preloads into it)
set -u
tag=$1; shift
out=gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "WRITE_SIZE GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  # (FETCH_SIZE and WRITE_SIZE do not fit one pass: "exceeds the capabilities of the hardware", and the aborted
  # profiler then sits there until the caller's limit: every pass under its own timeout)
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- "$@" > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "voxelize_kernel" not in k and "voxelize_narrow" not in k: continue
    print(k)
    m = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in d.items()}  # second half of the dispatches: warmed up
    for c in sorted(m): print(f"   {c:24s} n={len(d[c]):4d} mean={m[c]:.6g}")
    if "SQ_WAVES" in m and "SQ_LDS_IDX_ACTIVE" in m:
        W = m["SQ_WAVES"]
        print(f"   per wave: LDS instructions {m['SQ_INSTS_LDS'] / W:.0f}, LDS array cycles {m['SQ_LDS_IDX_ACTIVE'] / W:.0f} "
              f"(bank conflicts {m['SQ_LDS_BANK_CONFLICT'] / W:.0f}), VMEM rd {m['SQ_INSTS_VMEM_RD'] / W:.1f} wr {m['SQ_INSTS_VMEM_WR'] / W:.1f}, SMEM {m['SQ_INSTS_SMEM'] / W:.1f}")
        print(f"   LDS array cycles per compute unit (256 CUs): {m['SQ_LDS_IDX_ACTIVE'] / 256:.4g}")
    if "GRBM_GUI_ACTIVE" in m:
        print(f"   kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs): {m['GRBM_GUI_ACTIVE'] / 8:.4g}")
    if "SQ_WAVES" in m and "SQ_WAVE_CYCLES" in m:
        print(f"   wave life (SQ_WAVE_CYCLES/SQ_WAVES)        {m['SQ_WAVE_CYCLES'] / m['SQ_WAVES']:.0f}")
        print(f"   waiting share (SQ_WAIT_ANY/SQ_WAVE_CYCLES) {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
        print(f"   busy (SQ_BUSY_CYCLES)                      {m['SQ_BUSY_CYCLES']:.4g}; waves in flight per SQ-busy cycle {m['SQ_WAVE_CYCLES'] / m['SQ_BUSY_CYCLES']:.1f}")
        print(f"   VALU per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}  SALU per wave {m['SQ_INSTS_SALU'] / m['SQ_WAVES']:.0f}")
PY
