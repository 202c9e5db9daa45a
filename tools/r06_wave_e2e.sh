#!/bin/bash
# round 6, after the wave-per-member parse: the full-size sample in one process (tools/e2e_bench.py, 7 runs each) at device
# shares 50 / 75 / 100 of the sequence slices' inflate work, and at 50 with the former one-launch kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; rm -rf $d
python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_wave_e2e.err
for setting in "50 3" "75 3" "100 3" "50 1" "60 3" "50 3"; do
  set -- $setting
  SVX_BAM_DEVICE_INFLATE=$1 SVX_INFLATE_KERNEL=$2 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 7 2>> gpurun_out/r06_wave_e2e.err | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('share $1 kernel $2: median %.3f s  runs %s  cpu %.2f s  host-only runs %s  vcf ok %s  members on device %s' % (m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s']), m['cpu_seconds']['total'], r.get('host_inflate_only_runs_total_s'), r.get('vcf_matches_real_reference_digest', r.get('vcf_equal')), m.get('bgzf_members_inflated_on_device')))"
done | tee gpurun_out/r06_wave_e2e.txt
