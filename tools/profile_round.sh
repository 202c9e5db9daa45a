#!/bin/bash
# Refresh the judged evidence on the GPU box: bench line (with its extras), rocprofv3 kernel stats of the same
# command, PMC passes (separate runs).  Output under gpurun_out/$1/ (default r02); copy what is wanted into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r02}; mkdir -p $out
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline --e2e-scale 0 > $out/bench_under_rocprof.json 2> $out/rocprof.err
cp $out/stats/*/s_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
if [ "$2" = "pmc" ]; then bash tools/pmc.sh $out/pmc > $out/pmc.txt 2>&1; fi
tail -c 3000 $out/bench.json
