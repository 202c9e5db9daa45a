#!/bin/bash
# round 6: the device leg's staging and decode pipelined in phases (SVX_BAM_LEG_PHASES: 1 = the former order, all payloads
# first; default 3 for a full-size call) — the full-size sample in one process, 7 runs a setting, interleaved twice
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_leg_phases.err
for rep in 1 2; do for ph in ${PHASES:-1 3 2 4 6}; do
  SVX_BAM_LEG_PHASES=$ph python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 7 2>> gpurun_out/r06_leg_phases.err | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('phases $ph: median %.3f s  runs %s  cpu %.2f s  vcf ok %s' % (m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s']), m['cpu_seconds']['total'], r.get('vcf_matches_real_reference_digest', r.get('vcf_equal'))))"
done; done | tee gpurun_out/r06_leg_phases.txt
SVX_BAM_DEBUG=1 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 2 2>&1 >/dev/null | grep "device leg" | head -4 | cut -c1-300
