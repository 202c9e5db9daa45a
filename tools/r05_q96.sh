#!/bin/bash
# signature queue of 64 against 96 entries per round: parity, the product operating points, the dense probe, the step
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/q96; mkdir -p $out
for v in default q96; do
  lib=svim_asm_amd/libsvx.so; [ "$v" = q96 ] && lib=build/libsvx_q96.so
  echo "== $v parity"
  SVX_LIB=$lib timeout 900 python3 -m pytest tests/test_gpu_cigar.py tests/test_gpu_collect.py -m gpu -x -q 2>&1 | tail -2
  SVX_LIB=$lib timeout 200 python3 tools/fuzz_cigar.py --seconds 60 --seed 96 2>&1 | tail -2
  echo "== $v operating points"
  for i in 1 2; do SVX_LIB=$lib timeout 300 python3 tools/collect_probe.py latency_case product_point 2>/dev/null | tee -a $out/points_$v.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items(): print('  ', k, {x: v[x] for x in v if x in ('us_per_call','kernel_us','frac','launch_us','ms_per_call','kernels_us')})"; done
  SVX_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/pp_$v -o pp -- python3 tools/collect_probe.py product_point > /dev/null 2>&1
  f=$(find $out/pp_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -d, -f1-5 | cut -c1-150
  echo "== $v dense probe"
  SVX_LIB=$lib timeout 300 python3 tools/dense_probe.py 2>&1 | tail -8
done
bash tools/r05_ab.sh default q96 default q96
