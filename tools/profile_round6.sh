#!/bin/bash
# Round-6 evidence on the GPU box.  Output under gpurun_out/r06/; what is judged is copied into profiles/r06_*.
#   bench.json                  the default line (all legs: e2e, e2e_sharded, e2e_samples, e2e_cohort, summary)
#   bench_driver_cmd.json       the driver's own command (5 + 20)
#   kernel_stats.csv            rocprofv3 --kernel-trace --stats of the headline command, no extras
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06; mkdir -p $out
timeout 2400 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_driver_cmd.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/stats -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
tail -c 1600 $out/bench.json
head -8 $out/kernel_stats.csv | cut -c1-200
