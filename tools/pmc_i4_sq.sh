#!/bin/bash
# usage: tools/pmc_i4_sq.sh [I]  -- SQ counters of the trace launch at max_interactions = I
I=${1:-4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_i${I}_sq; rm -rf $O; mkdir -p $O
[ -n "$CPM_VARIANT" ] && export CPM_LIB=build/variants/$CPM_VARIANT.so
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 tools/i4_time.py $I 5 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR --output-format csv -d $O/b -- python3 tools/i4_time.py $I 5 > $O/b.log 2>&1
python3 tools/pmc_summary.py $O trace_kernel
