# usage: tools/ws_variants.sh variant...  -- the workspace point (tools/ws_kernels.py: frame + per-kernel us) under builds of build/variants/
# (LD_PRELOAD: the host layer links libcpm_hip.so)
for v in "" "$@"; do
  echo "== ${v:-product}"
  if [ -n "$v" ]; then LD_PRELOAD=$PWD/build/variants/$v.so python tools/ws_kernels.py 100 2>&1 | grep -v amdgpu.ids; else python tools/ws_kernels.py 100 2>&1 | grep -v amdgpu.ids; fi
done
