#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv compactly (newest file under a dir)."""
import csv, glob, os, re, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
f = max(files, key=os.path.getmtime)
print("#", f)
for r in csv.DictReader(open(f)):
    n = r["Name"]
    m = re.search(r"([A-Za-z_0-9:]+(<[^(]*>)?)\(", n)
    n = (m.group(1) if m else n)[-48:]
    print(f"{n:48s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f} pct={r['Percentage']}")
