"""Per-call kernel timeline from a rocprofv3 --kernel-trace CSV: mean duration of every launch of a call and the idle gap
before it (end of the previous kernel -> start of this one), over the back-to-back calls of tools/single_calls.py.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/single_calls.py cfg5 300
    python3 tools/call_gaps.py DIR [launches per call, default 4]
"""
import csv
import glob
import sys

import numpy as np

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void mvx::", "").replace("mvx::", "")[:40] for r in rows]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
# the back-to-back section: calls 40 .. 240 of the first timed loop (skip warm-up; stop before the profiling pass)
first = next(i for i, n in enumerate(names) if n.startswith("prep_kernel"))
lo, hi = first + 40 * per, first + 240 * per
print(f"{len(rows)} dispatches; calls 40..240 of the run, {per} launches per call")
tot = 0.0
for k in range(per):
    idx = np.arange(lo + k, hi, per)
    dur = (en[idx] - st[idx]) / 1e3
    gap = (st[idx] - en[idx - 1]) / 1e3
    tot += dur.mean() + gap.mean()
    print(f"{names[idx[0]]:42s} gap before {gap.mean():6.2f} us   duration {dur.mean():6.2f} us (p10 {np.percentile(dur, 10):.2f} p90 {np.percentile(dur, 90):.2f})")
idx = np.arange(lo, hi, per)
print(f"call period (start to start): {np.diff(st[idx]).mean() / 1e3:.2f} us; sum of gaps + durations {tot:.2f} us")
