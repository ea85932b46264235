#!/bin/bash
# usage: tools/profile_round.sh <tag>  -- on the GPU box: kernel-trace stats of bench.py, then two PMC passes
# (FETCH_SIZE, WRITE_SIZE; each in its own run with --kernel-trace only), summaries under gpurun_out/.
# The profiled runs pass --streams 0: bench.py's extra 'pipelined' figure runs 4 frames concurrently, which would
# inflate the per-kernel averages; with it off the kernel_stats averages are those of bench.py's own profile pass.
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/round_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 50 --warmup 5 --streams 0 > $O/bench_profiled.log 2>&1
python3 tools/kstats.py $O/stats > $O/kernel_stats_summary.txt; head -14 $O/kernel_stats_summary.txt
cp $(ls -t $(find $O/stats -name '*kernel_stats.csv') | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 0 > $O/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
python3 bench.py --steps 200 --warmup 20 > $O/bench.log 2>&1; grep '"metric"' $O/bench.log > $O/bench.json; cut -c1-400 $O/bench.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
