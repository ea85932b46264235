# usage: tools/variants_fast.sh workload variant...  -- the tolerance-mode frame of tools/fast_time.py under builds of build/variants/
WL=$1; shift
for v in "$@"; do
  echo "== $WL ${v:-product}"
  if [ -n "$v" ]; then export CPM_LIB=build/variants/$v.so; else unset CPM_LIB; fi
  python tools/fast_time.py $WL 2>&1 | grep -v amdgpu.ids | awk '/^fast/{p=1} p' | head -5
done
