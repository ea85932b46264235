#!/bin/bash
# usage: tools/prof_correlated.sh <tag>  -- per-kernel breakdown of the correlated update through the C++ processors
# (configs 3 and 5): rocprofv3 --kernel-trace --stats of tools/host_update_only.py; summaries under gpurun_out/
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for W in config3 config5; do
  O=gpurun_out/profc_${TAG}_$W; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/host_update_only.py $W 60 > gpurun_out/profc_${TAG}_$W.log 2>&1
  python3 tools/kstats.py $O | head -24
  cp $(ls -t $(find $O -name '*kernel_stats.csv') | head -1) gpurun_out/profc_${TAG}_${W}_kernel_stats.csv
  tail -2 gpurun_out/profc_${TAG}_$W.log | cut -c1-400
done
