#!/bin/bash
# usage: tools/build_variant.sh <name> [-DCPM_...=... ...]  -- another build of libcpm_hip.so under build/variants/ (git-ignored, travels
# to the GPU box); select it with CPM_LIB=build/variants/<name>.so (binding.load_library)
set -e
cd "$(dirname "$0")/.."
P=$(ls -d correlated*_amd)
NAME=$1; shift
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wall -Wno-unused-function -ldl "$@" -I include -I $P/csrc \
  -o build/variants/$NAME.so $P/csrc/cpm_core.hip $P/csrc/cpm_rng_emission.hip $P/csrc/cpm_trace.hip $P/csrc/cpm_sort.hip \
  $P/csrc/cpm_lightvolume.hip $P/csrc/cpm_fastvolume.hip $P/csrc/cpm_correlated.hip $P/csrc/cpm_temporal.hip $P/csrc/cpm_comm.hip $P/csrc/cpm_gl.hip
echo built build/variants/$NAME.so
