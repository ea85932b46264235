#!/bin/bash
# usage: tools/pmc_i4.sh [I]  -- on the GPU box: HBM-side read traffic (FETCH_SIZE) and L2 hit / miss of the trace launch at max_interactions = I
I=${1:-4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_i$I; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/i4_time.py $I 5 > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/t -- python3 tools/i4_time.py $I 5 > $O/t.log 2>&1
python3 - <<PY
import csv, glob, collections
for d, names in (("$O/f", ["FETCH_SIZE"]), ("$O/t", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"])):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in names:
        v = acc.get(n, [])
        if v: print(f"I = $I trace_kernel {n}: mean {sum(v) / len(v):.0f} over {len(v)} launches" + (f"  = {2 * 1024 * sum(v) / len(v) / 1e6:.1f} MB read (x2 gfx950 correction)" if n == "FETCH_SIZE" else ""))
PY
rm -rf $O/f $O/t
