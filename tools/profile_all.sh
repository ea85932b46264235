#!/bin/bash
# usage: tools/profile_all.sh <tag>  -- every artefact of a round in one call on the GPU box: profile_round.sh for config 2 and config 4's
# size, the correlated updates (configs 3 and 5) and the workspace point through the C++ processors (kernel stats + SQ counters).
# Copy what is to be judged from gpurun_out/ into profiles/ afterwards (tools/README.md).
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG config2 > gpurun_out/profile_round_$TAG.log 2>&1; echo "config2 done"
bash tools/profile_round.sh ${TAG}_config4 config4 > gpurun_out/profile_round_${TAG}_config4.log 2>&1; echo "config4 done"
bash tools/prof_correlated.sh $TAG > gpurun_out/prof_correlated_$TAG.log 2>&1; echo "correlated done"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/profw_$TAG; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/workspace_only.py 60 > gpurun_out/profw_$TAG.log 2>&1
cp $(ls -t $(find $O -name '*kernel_stats.csv') | head -1) gpurun_out/profw_${TAG}_kernel_stats.csv
bash tools/pmc_workspace.sh $TAG > gpurun_out/pmcw_${TAG}_summary.txt 2>&1; echo "workspace done"
