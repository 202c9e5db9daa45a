#!/bin/bash
# round 6: the product and the device decoder on files written the way an htslib built with libdeflate writes them (the
# synthetic writer at level 106 = libdeflate level 6) and by zlib at level 6
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lv in 106 6; do
  d=/tmp/svx_ds_lv$lv; rm -rf $d
  python3 tools/e2e_bench.py --scale 1.0 --bam-level $lv --keep $d --ranks "" --repeat 7 2> gpurun_out/r06_libdeflate.err | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); m=r['median_run']
print('BAMs at level $lv (%d / %d bytes): in one process median %.3f s  runs %s  cpu %.2f s  host-only %s  vcf ok %s' % (r['bam_bytes'][0], r['bam_bytes'][1], m['product_total_s'], ' '.join('%.3f' % x for x in r['all_runs_total_s']), m['cpu_seconds']['total'], [round(x,3) for x in r.get('host_inflate_only_runs_total_s', [])], r.get('vcf_matches_real_reference_digest', r.get('vcf_equal'))))"
  python3 tools/gpu_inflate_probe.py --scale 1.0 --dataset $d --members 30000 --min-payload 8192 --counts 7261,14000 2>/dev/null | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print('  device decoder on its SEQ members:', r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ms; compressed MB', round(r['compressed_bytes']/1e6,1), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
done | tee gpurun_out/r06_libdeflate.txt
