#!/bin/bash
# interleaved A/B with medians: tools/abmed.sh REPS "bench args" lib1.so lib2.so ...
reps=$1; args=$2; shift 2
mkdir -p gpurun_out; rm -f gpurun_out/abmed_*.txt
for r in $(seq $reps); do
  for lib in "$@"; do
    SVX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print(rf['kernel_ms'], rf['path_ms'], r['ms_per_step'])" >> gpurun_out/abmed_$(basename $lib .so).txt
  done
done
python - "$@" <<'PY'
import sys, numpy as np, os
for lib in sys.argv[1:]:
    a = np.loadtxt("gpurun_out/abmed_%s.txt" % os.path.basename(lib)[:-3]).reshape(-1, 3)
    print("%-34s kernel_ms median %.4f (min %.4f max %.4f)  path %.4f  step %.4f  n=%d" % (lib, np.median(a[:,0]), a[:,0].min(), a[:,0].max(), np.median(a[:,1]), np.median(a[:,2]), len(a)))
PY
