#!/bin/bash
# round 5, the last tree: the -m gpu suite, the upload probe, the bench line and the driver's command
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05e; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=5 > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 1800 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 200 $out/bench.json; echo
for i in 1 2; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_driver_cmd_$i.json 2>> $out/bench.err; python3 -c "
import json; r=json.loads(open('$out/bench_driver_cmd_$i.json').read().strip().splitlines()[-1]); rf=r['roofline']
print('driver cmd $i: value %.4g ms/step %.4f median5 %.4g kernel_ms %.4f frac %.3f path_frac %.3f step_frac %.3f throttled %s' % (r['value'], r['ms_per_step'], r['value_median_of_5'], rf['kernel_ms'], rf['frac'], rf['path_frac'], rf['step_frac'], r['host_throttled_ms_in_timed_region']))"; done
d=/tmp/svx_up_ds
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
timeout 600 python3 tools/upload_probe.py --dataset $d --repeat 7 > $out/upload_probe.json 2> $out/probe.err; cat $out/upload_probe.json
