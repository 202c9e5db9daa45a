#!/bin/bash
# Refresh the judged evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command,
# PMC passes (separate runs).  Output under gpurun_out/r01b/; copy what is wanted into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r01b; mkdir -p $out
timeout 600 python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
bash tools/pmc.sh $out/pmc > $out/pmc.txt 2>&1
tail -c 1500 $out/bench.json
