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
STEPS=${STEPS:-40}
WARMUP=${WARMUP:-20}
PREWARM=$(python3 -c "import bench; print(bench.PREWARM_LAUNCHES)")
# explicit --steps / --warmup: the slices below are derived from these, never from bench.py's defaults.
# Dispatch order of one run: PREWARM pre-warm + WARMUP warm-up + STEPS timed (pass 1, `value`) + 2 + STEPS (pass 2, events)
BENCH="python3 bench.py --cpu-seconds 0 --pmc-traffic off --batch $BATCH --steps $STEPS --warmup $WARMUP"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kernel_trace" -- $BENCH > "$out/kernel_trace.log" 2>&1 || echo "kernel-trace failed"
i=0
for grp in "WRITE_SIZE GRBM_GUI_ACTIVE" "FETCH_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -- $BENCH > "$out/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" "$tag" "$BATCH" "$STEPS" "$WARMUP" "$PREWARM" <<'PY'
import csv, glob, json, sys, collections, os
out, tag, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps, warmup, prewarm = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
calls_per_run = prewarm + warmup + steps + 2 + steps  # bench.py steps per run (each = launches_per_step dispatches)
# kernel stats
stats = []
for f in glob.glob(out + "/kernel_trace/**/*kernel_stats.csv", recursive=True):
    stats += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = [f"# rocprofv3 --kernel-trace --stats (python3 bench.py --cpu-seconds 0 --batch {batch} --steps {steps} --warmup {warmup}: "
         f"{prewarm} pre-warm + {warmup} warm-up + {steps} timed steps (pass 1) + 2 + {steps} steps with events (pass 2))"]
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
        summary = {"workload": "cfg2", "batch": batch, "steps_profiled": calls_per_run, "kernel": k[:80], "write_bytes_per_launch": wr,
                   "fetch_bytes_per_launch_corrected": rd, "hbm_bytes_per_launch": wr + rd, "tag": tag,
                   "note": "WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reads half on gfx950)"}
        # per-dispatch durations, in dispatch order: pre-warm + warm-up, then pass 1 (the timed steps), then pass 2
        durs = []
        for f in glob.glob(out + "/kernel_trace/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "voxelize_kernel" in row["Kernel_Name"]:
                    durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        durs = [d for _, d in sorted(durs)]
        lps = max(1, len(durs) // calls_per_run) if durs else 1
        if durs and len(durs) == lps * calls_per_run:
            w0, t0, t1 = lps * (prewarm + warmup), lps * (prewarm + warmup), lps * (prewarm + warmup + steps)
            timed, pass2 = durs[t0:t1], durs[-lps * steps:]
            summary["rocprof_timed_steps_avg_kernel_ns"] = sum(timed) / len(timed)
            summary["rocprof_pass2_avg_kernel_ns"] = sum(pass2) / len(pass2)
            summary["rocprof_warmup_avg_kernel_ns"] = sum(durs[:w0]) / max(1, w0)
            lines.append(f"voxelize_kernel per-dispatch: pre-warm + warm-up launches avg {summary['rocprof_warmup_avg_kernel_ns']:.0f} ns, "
                         f"timed steps (pass 1, {len(timed)} launches) avg {summary['rocprof_timed_steps_avg_kernel_ns']:.0f} ns, min {min(timed)} max {max(timed)}; "
                         f"pass 2 (with events) avg {summary['rocprof_pass2_avg_kernel_ns']:.0f} ns")
        elif durs:
            lines.append(f"voxelize_kernel: {len(durs)} dispatches, not a multiple of the {calls_per_run} steps of one run - slices not derived")
        for r in stats:
            if "voxelize_kernel" in r["Name"]:
                summary["rocprof_avg_kernel_ns"] = float(r["AverageNs"])
                summary["launches_per_step"] = max(1, int(r["Calls"]) // calls_per_run)
                summary["molecules_per_launch"] = batch // max(1, int(r["Calls"]) // calls_per_run)
open(out + f"/summary_{tag}.txt", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(out + "/pmc_latest.json", "w"), indent=1)
print("\n".join(lines[:12]))
print(json.dumps(summary))
PY
