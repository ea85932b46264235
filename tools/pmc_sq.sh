#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [kernel-name filter]   -- SQ counters of the frame's kernels (two passes of <= 8 SQ counters)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcg_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 tools/gather_only.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 tools/gather_only.py > $O/b.log 2>&1
python3 tools/pmc_summary.py $O $2

