#!/bin/bash
# round 5: the device leg without spinning waits (8 staging threads, blocking events): kernel geometry variants, then the
# full-size run at shares 0 / 50 interleaved
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_devleg4; mkdir -p $out
bash tools/r05_infl_geom2.sh
timeout 600 python3 -m pytest tests/test_gpu_device_pool.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -2
d=/tmp/svx_cli_dataset
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
SVX_BAM_DEBUG=1 SVX_BAM_DEVICE_INFLATE=50 timeout 600 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 2 2>&1 >/dev/null | grep "device leg" | tail -4 | cut -c1-260
for i in 1 2 3; do for v in 0 50; do
  SVX_BAM_DEVICE_INFLATE=$v timeout 600 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "" --repeat 9 > $out/e2e_${v}_$i.json 2> $out/e2e.err
  python3 -c "
import json; r=json.loads(open('$out/e2e_${v}_$i.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('share $v: total %.4f all %s ok %s cpu median-run %.2f best-run %.2f dev %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], r.get('vcf_matches_real_reference_digest'), m['cpu_seconds']['total'], r['best_run']['cpu_seconds']['total'], r.get('bgzf_members_inflated_on_device')), {k2: round(v*1e3,1) for k2,v in m.get('collect_stages_s',{}).items() if k2 in ('load_s','sequences_wait_s')})"
done; done
