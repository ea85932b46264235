#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: mean per dispatch of each counter per kernel."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"([A-Za-z_0-9:]+(<[^(]*>)?)\(", n)
        n = (m.group(1) if m else n)[-40:]
        if pat and pat not in n: continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
