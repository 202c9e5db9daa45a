#!/bin/bash
# round 6: the record walks' share of the default check on the device leg (svx_bam_set_defer_verify) against checking in the
# walks (SVX_BAM_NO_DEFER_VERIFY=1): the full-size sample in one process (7 runs, twice), the fresh command (11 processes,
# twice), svim-asm-cohort N = 16
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_defer.err
{
for rep in 1 2; do for off in "" 1; do
  SVX_BAM_NO_DEFER_VERIFY=$off python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 7 2>> gpurun_out/r06_defer.err | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('in one process, deferral %s: median %.3f s  runs %s  cpu %.2f s (collect %.2f)  vcf ok %s  members on device %s' % ('off' if '$off' else 'on ', m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s']), m['cpu_seconds']['total'], m['cpu_seconds']['collect'], r.get('vcf_matches_real_reference_digest', r.get('vcf_equal')), m.get('bgzf_members_inflated_on_device')))"
done; done
for rep in 1 2; do for off in "" 1; do
  echo -n "command, deferral $([ -n "$off" ] && echo off || echo on ): "; python3 tools/cli_timeline.py $d 11 SVX_BAM_NO_DEFER_VERIFY=$off 2>&1 | grep -E "^wall-clock|^CPU seconds" | tr '\n' ' '; echo
done; done
} | tee gpurun_out/r06_defer.txt
for off in "" 1 "" 1; do
  SVX_BAM_NO_DEFER_VERIFY=$off python3 tools/r06_cohort_ab.py --dataset /tmp/svx_cohort_ds --n 16 --out gpurun_out/r06_defer_cohort_$off.jsonl --settings "4:100:400:1" > /dev/null 2>> gpurun_out/r06_defer.err
  python3 -c "
import json
for l in open('gpurun_out/r06_defer_cohort_$off.jsonl'):
    r=json.loads(l); print('cohort N=16, deferral %s: %.2f samples/s  wall %.2f s  cpu/sample %.2f  ok %s' % ('off' if '$off' else 'on ', r['samples_per_s'], r['wall_s'], r['cpu_seconds_per_sample'], all(x is not False for x in r['vcf_matches_real_reference_digest'])))"
done | tee -a gpurun_out/r06_defer.txt
