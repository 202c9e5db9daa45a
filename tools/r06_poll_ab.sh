#!/bin/bash
# A/B: variants of the library on the full-size sample in one process (7 runs each, interleaved twice): wall, CPU seconds, the sequence calls' CPU seconds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for rep in 1 2; do for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  SVX_LIB=$PWD/$lib python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 7 2>/dev/null | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('$v: median %.3f s  runs %s  cpu %.2f s  sequences cpu %.3f wait %.3f' % (m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s'][1:]), m['cpu_seconds']['total'], m.get('collect_stages_s',{}).get('sequences_cpu_s',-1), m.get('collect_stages_s',{}).get('sequences_wait_s',-1)))"
done; done
