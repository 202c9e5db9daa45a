#!/bin/bash
# A/B bench of several libsvx builds on the GPU box: tools/ab.sh lib1.so lib2.so ...
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  SVX_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('value %.1f Gops/s  ms/step %.4f  kernel_ms %.4f  path_ms %.4f  achieved %.0f GB/s frac %.3f' % (r['value']/1e9, r['ms_per_step'], rf['kernel_ms'], rf['path_ms'], rf['achieved'], rf['frac']))"
done
