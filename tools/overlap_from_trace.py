"""Kernel start / end stamps of a rocprofv3 --kernel-trace run of tools/root_overlap.py -> how much of the root's add launches' time ran while a
kernel of the frame stream was executing.  usage: python tools/overlap_from_trace.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
adds = [r for r in rows if "bricklist_add_kernel" in r[0] or "bricklist_index_kernel" in r[0]]
frame = [r for r in rows if any(k in r[0] for k in ("trace_kernel", "fast_count_kernel", "fast_scatter_kernel", "fast_brick_kernel"))]
adds = adds[16:]   # (the first frames' launches carry one-off costs)
t_add = sum(e - s for _, s, e in adds)
ov = 0
for _, s, e in adds:
    for _, fs, fe in frame:
        if fe <= s:
            continue
        if fs >= e:
            break
        ov += min(e, fe) - max(s, fs)
# the frame stream's own pace: start of one trace to the start of the next
tr = [s for n, s, e in frame if "trace_kernel" in n]
gaps = sorted(b - a for a, b in zip(tr[8:], tr[9:]))
per = {}
for name, s, e in rows:
    import re
    m = re.search(r"([A-Za-z_0-9]+_kernel)", name)
    k = m.group(1) if m else name[:40]
    per.setdefault(k, []).append((e - s) / 1e3)
out = {"add_launches": len(adds), "add_time_us_total": round(t_add / 1e3, 1), "of_which_beside_a_frame_kernel_us": round(ov / 1e3, 1),
       "overlapped_fraction": round(ov / max(t_add, 1), 3),
       "frame_period_us_median": round(gaps[len(gaps) // 2] / 1e3, 1) if gaps else None,
       "mean_us": {k: round(sum(v) / len(v), 2) for k, v in per.items() if len(v) >= 8},
       "note": "fraction of the bricklist_index_kernel + bricklist_add_kernel execution time during which a kernel of the frame stream (trace / count / scatter / brick gather) "
               "was executing too, from rocprofv3's kernel start / end stamps; one GPU, two HIP streams, the adds behind an event after each gather; frame_period = "
               "start of a trace launch to the start of the next (the root's own frames with the adds beside them)"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
