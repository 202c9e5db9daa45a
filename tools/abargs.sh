#!/bin/bash
# bench one library with several argument sets: tools/abargs.sh lib.so "args1" "args2" ...
lib=$1; shift
for a in "$@"; do
  echo "== $a"
  SVX_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('value %.1f Gops/s  ms/step %.4f  kernel_ms %.4f  path_ms %.4f  achieved %.0f GB/s frac %.3f' % (r['value']/1e9, r['ms_per_step'], rf['kernel_ms'], rf['path_ms'], rf['achieved'], rf['frac']))"
done
