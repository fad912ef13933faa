"""Idle gaps between the launches of the timed bench steps, from a rocprofv3 --kernel-trace CSV of `python3 bench.py`:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --cpu-seconds 0
    python3 tools/bench_gaps.py DIR [launches per step, default 4] [timed steps, default 40]
"""
import csv
import glob
import sys

import numpy as np

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "mvx" in r["Kernel_Name"]][-per * steps:]
st = np.array([int(r["Start_Timestamp"]) for r in rows])
en = np.array([int(r["End_Timestamp"]) for r in rows])
tot = 0.0
for k in range(per):
    idx = np.arange(k, per * steps, per)
    idx = idx[idx > 0]
    gap, dur = ((st[idx] - en[idx - 1]) / 1e3).mean(), ((en[idx] - st[idx]) / 1e3).mean()
    tot += gap + dur
    print(f'{rows[k]["Kernel_Name"].replace("void ", "").replace("mvx::", "")[:44]:44s} gap before {gap:6.2f} us   duration {dur:8.2f} us')
print(f"step period (start to start) {np.diff(st[0::per]).mean() / 1e3:.2f} us; gaps + durations {tot:.2f} us")
