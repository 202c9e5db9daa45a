#!/bin/bash
# round 6: BASELINE config 5 as a diploid sample (1 200-member sequence-slice calls) with the device leg's minimum at 3000
# (the calls stay on the host), 500 and 100, in one process, 7 runs each, interleaved twice
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_c5_ds; rm -rf $d
python3 tools/e2e_bench.py --config5 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_wave_min.err
for rep in 1 2; do for m in 3000 500 100; do
  SVX_BAM_DEVICE_INFLATE_MIN=$m python3 tools/e2e_bench.py --config5 --dataset $d --ranks "" --repeat 7 2>> gpurun_out/r06_wave_min.err | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('min $m: median %.3f s  runs %s  cpu %.2f s  vcf ok %s  members on device %s' % (m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s']), m['cpu_seconds']['total'], r.get('vcf_matches_real_reference_digest', r.get('vcf_equal')), m.get('bgzf_members_inflated_on_device')))"
done; done | tee gpurun_out/r06_wave_min.txt
