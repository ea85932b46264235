#!/bin/bash
# usage: tools/r2_check.sh <pytest-targets...>   -- GPU box: given tests, then the stage timing of both formulations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15
python tools/fast_time.py config2 50 2>&1 | tail -40
