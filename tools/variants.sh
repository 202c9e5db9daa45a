#!/bin/bash
# the bench headline on other workloads, same binary, same box: tools/variants.sh > profiles/rNN_workload_variants.txt
cd "$GRAFT_REPO_ROOT"
run() {
  python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']; c=r['config']
print('%-44s ops/step %12d  step %8.1f us  value %7.1f G ops/s  k_cigar_tiles %7.1f us  %6.0f GB/s (frac %.3f)  path %7.1f us (frac %.3f)' % (' '.join(sys.argv[1:]) or '(default: config 2 x 256, packed)', c['ops_per_step_per_gpu'], r['ms_per_step']*1e3, r['value']/1e9, rf['kernel_ms']*1e3, rf['achieved'], rf['frac'], rf['path_ms']*1e3, rf['path_frac']))" -- "$@"
}
run --samples 1
run --samples 64
run
run --pipeline
run --min-sv-size 100000000
run --config 5 --samples 64
run --layout soa
