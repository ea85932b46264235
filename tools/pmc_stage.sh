#!/bin/bash
# usage: tools/pmc_stage.sh <tag> <stage> [kernel filter]  -- SQ counters (two passes of <= 8) of one stage's kernels
TAG=${1:-x}; STAGE=${2:-trace}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcs_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 tools/stage_only.py $STAGE 10 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 tools/stage_only.py $STAGE 10 > $O/b.log 2>&1
python3 tools/pmc_summary.py $O $3 | tee $O/summary.txt
