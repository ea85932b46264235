#!/bin/bash
# usage: tools/pmc_tcp.sh <tag> <stage> [kernel filter]  -- texture-path counters (TA / TCP / TD) of one stage's kernels
TAG=${1:-x}; STAGE=${2:-trace}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmct_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $O/a -- python3 tools/stage_only.py $STAGE 10 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d $O/b -- python3 tools/stage_only.py $STAGE 10 > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/c -- python3 tools/stage_only.py $STAGE 10 > $O/c.log 2>&1
python3 tools/pmc_summary.py $O $3 | tee $O/summary.txt
