#!/bin/bash
# round 5: the whole -m gpu suite, then the evidence of the final tree (tools/profile_round5.sh, both parts)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.txt 2>&1; tail -14 $out/pytest_gpu.txt
bash tools/profile_round5.sh r05 2>&1 | tail -5
bash tools/profile_round5.sh r05 pmc 2>&1 | tail -8
# (the bench line of the final tree: tools/r05_bench.sh; config 5 + fuzz campaigns: tools/r05_last.sh)
