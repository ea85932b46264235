#!/bin/bash
# usage: tools/gpu_check.sh <tag> [pytest-args]   -- run GPU parity tests + rocprof'd bench on the GPU box
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
grep '"metric"' gpurun_out/bench_$TAG.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'ms',d['ms_per_step']); print(d['frame']['stage_ms'])" || tail -20 gpurun_out/bench_$TAG.log
python3 tools/kstats.py gpurun_out/prof_$TAG | head -14
