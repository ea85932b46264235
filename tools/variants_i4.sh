# usage: tools/variants_i4.sh I variant...  -- tools/i4_time.py under builds of build/variants/
I=$1; shift
for v in "$@"; do
  echo "== I=$I ${v:-product}"
  if [ -n "$v" ]; then export CPM_LIB=build/variants/$v.so; else unset CPM_LIB; fi
  timeout -k 10 120 python tools/i4_time.py $I 40 2>&1 | grep -v amdgpu.ids | head -2
done
