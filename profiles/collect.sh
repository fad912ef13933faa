#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag>          e.g. profiles/collect.sh r01
# 1. rocprofv3 --kernel-trace --stats of the exact bench command -> per-kernel average durations
# 2. separate --pmc passes (WRITE_SIZE ; FETCH_SIZE ; SQ counters) of the same command
# 3. summary json with HBM bytes per voxelize launch (gfx950 correction: FETCH_SIZE x 2), used by bench.py `traffic`
# Outputs land in gpurun_out/profiles_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
tag=${1:-r01}
out=gpurun_out/profiles_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
BATCH=${BATCH:-256}
BENCH="python3 bench.py --cpu-seconds 0 --batch $BATCH"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kernel_trace" -- $BENCH > "$out/kernel_trace.log" 2>&1 || echo "kernel-trace failed"
i=0
for grp in "WRITE_SIZE GRBM_GUI_ACTIVE" "FETCH_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -- $BENCH > "$out/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" "$tag" "$BATCH" <<'PY'
import csv, glob, json, sys, collections, os
out, tag, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
# kernel stats
stats = []
for f in glob.glob(out + "/kernel_trace/**/*kernel_stats.csv", recursive=True):
    stats += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = [f"# rocprofv3 --kernel-trace --stats (python3 bench.py --cpu-seconds 0 --batch {batch}, defaults: 20 warm-up + 40 timed steps)"]
for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"])):
    lines.append(f'{r["Name"][:110]:110s} calls={r["Calls"]:>4s} avg_ns={float(r["AverageNs"]):12.1f} pct={r["Percentage"]}')
lines.append("")
lines.append("# rocprofv3 --pmc passes (mean per dispatch)")
summary = {}
for k, d in agg.items():
    lines.append(k[:160])
    for c, v in sorted(d.items()):
        lines.append(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
    if "voxelize_kernel" in k and "WRITE_SIZE" in d:
        wr = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024.0          # KB -> bytes (exact for 16-B/lane stores)
        rd = 2.0 * sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [0]))) * 1024.0  # gfx950: x2
        summary = {"workload": "cfg2", "batch": batch, "steps_profiled": 60, "kernel": k[:80], "write_bytes_per_launch": wr,
                   "fetch_bytes_per_launch_corrected": rd, "hbm_bytes_per_launch": wr + rd, "tag": tag,
                   "note": "WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reads half on gfx950)"}
        # per-dispatch durations: the last 40 launches are bench.py's timed steps (the 20 before them warm the clock up)
        durs = []
        for f in glob.glob(out + "/kernel_trace/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "voxelize_kernel" in row["Kernel_Name"]:
                    durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        durs = [d for _, d in sorted(durs)]
        if durs:
            timed = durs[-40:]
            summary["rocprof_timed_steps_avg_kernel_ns"] = sum(timed) / len(timed)
            summary["rocprof_warmup_avg_kernel_ns"] = sum(durs[:-40]) / max(1, len(durs[:-40]))
            lines.append(f"voxelize_kernel per-dispatch: warm-up launches avg {summary['rocprof_warmup_avg_kernel_ns']:.0f} ns, "
                         f"timed steps (last 40) avg {summary['rocprof_timed_steps_avg_kernel_ns']:.0f} ns, min {min(timed)} max {max(timed)}")
        for r in stats:
            if "voxelize_kernel" in r["Name"]:
                summary["rocprof_avg_kernel_ns"] = float(r["AverageNs"])
                summary["launches_per_step"] = int(r["Calls"]) // 60
                summary["molecules_per_launch"] = batch // max(1, int(r["Calls"]) // 60)
open(out + f"/summary_{tag}.txt", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(out + "/pmc_latest.json", "w"), indent=1)
print("\n".join(lines[:12]))
print(json.dumps(summary))
PY
