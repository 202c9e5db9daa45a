#!/bin/bash
# round 5, last tree: the command line as it ships (no device leg, no inflate lanes) against SVX_BAM_DEVICE_INFLATE=50,
# one process and four rank processes; the device-pool tests
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_cli_ab2; mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_device_pool.py tests/test_large_golden.py -x -q -m gpu 2>&1 | tail -2
d=/tmp/svx_cli_dataset
[ -f $d/hap1.bam ] || timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for i in 1 2 3; do for v in default 50; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=$v; fi
  echo "== command line, SVX_BAM_DEVICE_INFLATE=$v"; python3 tools/cli_timeline.py $d 5 | sed 's/ | +[0-9.]* STEP 3.*os._exit//'
done; done
for i in 1 2; do for v in default 50; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=$v; fi
  timeout 600 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "1,4" --no-in-process > $out/sharded_${v}_$i.json 2> $out/err.txt
  python3 -c "
import json; r=json.loads(open('$out/sharded_${v}_$i.json').read().strip().splitlines()[-1]); print('ranks $v:', [(x['ranks'], round(x['wall_s'],3), [round(y,3) for y in x.get('all_wall_s',[])]) for x in r.get('cli_ranks',[])])"
done; done
