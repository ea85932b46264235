#!/bin/bash
# usage: tools/pmc_workspace.sh <tag>  -- SQ counters of the workspace point's kernels (the two-light network through the C++ processors)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcw_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 tools/workspace_only.py 10 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 tools/workspace_only.py 10 > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --output-format csv -d $O/c -- python3 tools/workspace_only.py 10 > $O/c.log 2>&1
python3 tools/pmc_summary.py $O fast_halo
