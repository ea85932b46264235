#!/bin/bash
# usage: tools/prof_correlated.sh <tag>  -- kernel trace of the correlated-update benchmark (configs 3 and 5)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/profc_$TAG; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/bench_correlated.py > gpurun_out/benchc_$TAG.log 2>&1
python3 tools/kstats.py $O | head -40
tail -1 gpurun_out/benchc_$TAG.log | cut -c1-600
