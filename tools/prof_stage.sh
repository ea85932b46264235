#!/bin/bash
# usage: tools/prof_stage.sh <tag> <stage> [repeats]  -- rocprofv3 kernel-trace stats of one stage (tools/stage_only.py)
TAG=${1:-x}; STAGE=${2:-frame_fast}; REPS=${3:-30}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/stage_only.py $STAGE $REPS > $O/run.log 2>&1
python3 tools/kstats.py $O | tee $O/summary.txt | head -24
