#!/bin/bash
# round 5: the device leg's minimum member count (svx_bam_set_device_inflate_min): tests, config 5 (a small sample: the
# leg must stay off) with the default and with the limit at 0, the full-size run once more
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_min; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_device_pool.py tests/test_large_golden.py tests/test_config5_golden.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
for v in default 0; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=0; fi
  timeout 900 python3 tools/e2e_bench.py --config5 --repeat 7 --ranks "" > $out/e2e_config5_$v.json 2> $out/c5.err
  python3 -c "
import json; r=json.loads(open('$out/e2e_config5_$v.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('config5 $v: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s dev %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest'), r.get('bgzf_members_inflated_on_device')), {k2: round(v*1e3,1) for k2,v in m.get('collect_stages_s',{}).items() if k2 in ('load_s','sequences_wait_s')})"
done
unset SVX_BAM_DEVICE_INFLATE
d=/tmp/svx_min_ds
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 5 > $out/e2e_full.json 2> $out/full.err
python3 -c "
import json; r=json.loads(open('$out/e2e_full.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('full: total %.4f all %s host-only %s ok %s cpu %.2f dev %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], r.get('host_inflate_only_runs_total_s'), r.get('vcf_matches_real_reference_digest'), m['cpu_seconds']['total'], r.get('bgzf_members_inflated_on_device')))"
