"""Print per-kernel average durations from a rocprofv3 --stats output directory (python3 tools/kstats.py DIR)."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:48]:48s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f}")
