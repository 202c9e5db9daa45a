#!/bin/bash
# Refresh the judged evidence on the GPU box.  Output under gpurun_out/$1/ (default r02); copy what is wanted
# into profiles/.
#   bench.json                     the default bench line (all legs)
#   kernel_stats.csv               rocprofv3 --kernel-trace --stats of the headline command (no extras: the
#                                  averages are those of the 256-sample cohort launches only)
#   kernel_stats_extras.csv        same for the latency / pair / edit-distance legs (tools/kbench.py)
#   pmc/ (with "pmc" as $2)        counter passes, separate runs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03}; mkdir -p $out
timeout 1500 python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/stats -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_extras -o s -- python3 tools/kbench.py latency pair editdist > $out/kbench_under_rocprof.json 2>> $out/rocprof.err
find $out/stats_extras -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats_extras.csv \;
rm -rf $out/stats $out/stats_extras
if [ "$2" = "pmc" ]; then bash tools/pmc.sh $out/pmc > $out/pmc.txt 2>&1; rm -f $out/pmc/pass*_kernel_trace.csv; fi
tail -c 600 $out/bench.json
