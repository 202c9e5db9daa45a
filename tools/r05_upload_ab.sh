#!/bin/bash
# round 5: the full-size run (a) as built: pool allocated beside the walk's last pieces + uploaded by the reader,
# (b) without the reader's upload (SVX_BAM_DEVICE_POOL=0), (c) neither (+ SVX_BAM_LATE_POOL=1: the tree before) — interleaved on one box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_upload; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_device_pool.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest_pool.txt 2>&1; tail -3 $out/pytest_pool.txt
d=/tmp/svx_cli_dataset
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for i in 1 2 3; do for v in a b c; do
  unset SVX_BAM_DEVICE_POOL SVX_BAM_LATE_POOL
  [ $v = b ] && export SVX_BAM_DEVICE_POOL=0
  [ $v = c ] && export SVX_BAM_DEVICE_POOL=0 SVX_BAM_LATE_POOL=1
  timeout 600 python3 tools/e2e_bench.py --dataset $d --ranks "" --repeat 7 > $out/e2e_${v}_$i.json 2> $out/e2e.err
  python3 -c "
import json; r=json.loads(open('$out/e2e_${v}_$i.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('$v: total %.4f all %s collect %.4f pair %.4f vcf %.4f' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s']), {k2: round(v*1e3,2) for k2,v in m.get('collect_stages_s',{}).items() if not k2.endswith('cpu_s')})"
done; done
unset SVX_BAM_DEVICE_POOL SVX_BAM_LATE_POOL
timeout 600 python3 tools/upload_probe.py --dataset $d --repeat 7 > $out/upload_probe.json 2> $out/probe.err; cat $out/upload_probe.json; tail -3 $out/probe.err
python3 tools/cli_timeline.py $d 5
