"""Kernel timeline (rocprofv3 --kernel-trace csv) of a few steady-state steps: python3 tools/overlap_trace.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "mvx" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = rows[len(rows) // 2: len(rows) // 2 + 16]
base = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    name = r["Kernel_Name"].split("(")[0].replace("void mvx::", "")[:28]
    print(f"{name:30s} queue {r.get('Queue_Id', '?'):>3s} start {s / 1e3:9.1f} us  end {e / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}")
