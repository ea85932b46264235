#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [workload]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]":
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE tallies 128-B read
requests at 64 B, so the read side is doubled; WRITE_SIZE is taken as is.  Infinity-Cache
hits are included (the counters sit on the L2's fabric side), so "traffic" is L2<->fabric
bytes: an upper bound of the HBM bytes.
"""
import csv, glob, json, os, re, sys, collections


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", n)
            acc[m.group(1) if m else n].append(float(r["Counter_Value"]))
    return acc


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "config2",
       "unit": "bytes per launch (mean over dispatches)",
       "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 128-B requests); WRITE_SIZE KiB x 1024",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [])
    w = write.get(k, [])
    fb = 2 * 1024 * sum(f) / len(f) if f else None
    wb = 1024 * sum(w) / len(w) if w else None
    out["kernels"][k] = {"read_bytes": fb, "write_bytes": wb,
                         "traffic_bytes": (fb or 0) + (wb or 0), "dispatches": max(len(f), len(w))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"])[:14]:
    print(f"{k:40s} read {v['read_bytes'] or 0:14.0f}  write {v['write_bytes'] or 0:14.0f}  n={v['dispatches']}")
