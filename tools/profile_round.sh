#!/bin/bash
# usage: tools/profile_round.sh <tag> [workload]  -- on the GPU box: kernel-trace stats of bench.py, two PMC traffic passes
# (FETCH_SIZE, WRITE_SIZE; each in its own run with --kernel-trace only), SQ counters of the frame's kernels, then the
# bench line itself.  Summaries under gpurun_out/round_<tag>/ (copy what is to be judged into profiles/).
# The profiled runs pass --no-extras: the extra figures (other formulations, concurrent frames, ...) would mix other
# kernels and concurrent launches into the per-kernel averages.
TAG=${1:-x}
WL=${2:-config2}   # bench.py --workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/round_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $WL --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_profiled.log 2>&1
python3 tools/kstats.py $O/stats > $O/kernel_stats_summary.txt; head -14 $O/kernel_stats_summary.txt
cp $(ls -t $(find $O/stats -name '*kernel_stats.csv') | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $WL
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_a -- python3 tools/stage_only.py frame_fast 10 $WL > $O/sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq_b -- python3 tools/stage_only.py frame_fast 10 $WL > $O/sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/sq_c -- python3 tools/stage_only.py frame_fast 10 $WL > $O/sq_c.log 2>&1
mkdir -p $O/sq; cp -r $O/sq_a $O/sq/a; cp -r $O/sq_b $O/sq/b; cp -r $O/sq_c $O/sq/c
python3 tools/pmc_summary.py $O/sq > $O/sq_counters_summary.txt
python3 bench.py --workload $WL --steps 200 --warmup 20 > $O/bench.log 2>&1; grep '"metric"' $O/bench.log > $O/bench.json; cut -c1-600 $O/bench.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/sq_a $O/sq_b $O/sq_c $O/sq
