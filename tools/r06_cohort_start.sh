#!/bin/bash
# round 6: what a cohort process's first groups cost (tools/cohort_timeline.py, N = 8): token arena size (variant libraries), lanes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/ds1; [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for rep in 1 2; do for v in "$@"; do
  lib=svim_asm_amd/libsvx.so; lanes=""; 
  case $v in default) ;; lanes2) lanes=2;; *) lib=build/libsvx_$v.so;; esac
  echo "== $v"; SVX_LIB=$PWD/$lib SVX_COHORT_LANES=$lanes python3 tools/cohort_timeline.py $d ${N:-8} 2>&1 | grep -E "^rc|COLLECT done|files closed|cohort-0" | cut -c1-330
done; done
