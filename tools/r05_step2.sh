#!/bin/bash
# round 5, step 2: the metric's own configuration under -m gpu; a two-rank bench line (nccl, then gloo) on the one device
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s2; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_full_golden.py -x -q -m gpu --durations=5 > $out/pytest_full.txt 2>&1
tail -12 $out/pytest_full.txt
timeout 600 python3 bench.py --gpus 2 --share-device --backend nccl --no-extras --no-cpu-baseline --init-timeout 60 > $out/bench_n2_nccl.json 2> $out/bench_n2_nccl.err
echo "nccl rc=$?"; tail -c 600 $out/bench_n2_nccl.err; tail -c 1500 $out/bench_n2_nccl.json
timeout 600 python3 bench.py --gpus 2 --share-device --backend gloo --no-extras --no-cpu-baseline > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err
echo "gloo rc=$?"; tail -c 300 $out/bench_n2_gloo.err; python3 -c "
import json; r=json.load(open('$out/bench_n2_gloo.json')); print({k: r[k] for k in ('value','n_gpus','ms_per_step','backend','collective_world_verified','distinct_devices','value_median_of_5')}); print(r['ranks'])"
