#!/bin/bash
# does the host's idle time in front of the warm-up steps (bench.py --settle-ms: one CPU-quota period has to roll over)
# change what the driver's 5 + 20 form measures?  interleaved, three runs each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do for ms in 300 120 0; do
  python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --settle-ms $ms 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print('settle $ms: ms_per_step %.4f  median_of_5 %.4f  sustained %.4f  host_throttled_ms %s  all %s' % (r['ms_per_step'], 395304576/r['value_median_of_5']*1e3 if r.get('value_median_of_5') else 0, r['sustained']['ms_per_step'] if r.get('sustained') else 0, r.get('host_throttled_ms_in_timed_region'), [round(x,4) for x in r['all_regions_ms_per_step']]))"
done; done
