#!/bin/bash
# usage: tools/variants_time.sh [workload] -- per-kernel times of the tolerance-mode frame for the default build and every build/variants/*.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=${1:-config2}
echo "== default"; python tools/fast_time.py $WL 50 2>&1 | grep -A8 "^fast"
for v in build/variants/*.so; do echo "== $v"; CPM_LIB=$v python tools/fast_time.py $WL 50 2>&1 | grep -A8 "^fast\|rror"; done
