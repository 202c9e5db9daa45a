#!/bin/bash
# round 5: the -m gpu suite and the bench line on the final tree
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05c; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 1800 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 200 $out/bench.json; echo
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_driver_cmd.json 2>> $out/bench.err
