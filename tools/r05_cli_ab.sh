#!/bin/bash
# round 5: the command line (one process, and four rank processes on one device) with and without the device leg
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_cli_ab; mkdir -p $out
d=/tmp/svx_cli_dataset
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for i in 1 2 3; do for v in default 0; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=0; fi
  echo "== command line, SVX_BAM_DEVICE_INFLATE=$v"; python3 tools/cli_timeline.py $d 5 | sed 's/ | +[0-9.]* STEP 3.*os._exit//' 
done; done
for i in 1 2; do for v in default 0; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=0; fi
  timeout 600 python3 tools/e2e_bench.py --scale 1.0 --dataset $d --ranks "2,4" --no-in-process > $out/sharded_${v}_$i.json 2> $out/err.txt
  python3 -c "
import json; r=json.loads(open('$out/sharded_${v}_$i.json').read().strip().splitlines()[-1]); print('sharded $v:', [(x['ranks'], round(x['wall_s'],3), [round(y,3) for y in x.get('all_wall_s',[])]) for x in r.get('cli_ranks',[])])"
done; done
