#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_leg_phases.err
for rep in 1 2 3; do for ph in ${PHASES:-1 2 3 4}; do SVX_BAM_LEG_PHASES=$ph python3 tools/r06_leg_probe.py $d 2>&1 | tail -1; done; done | tee gpurun_out/r06_leg_probe.txt
SVX_BAM_LEG_PHASES=1 SVX_INFLATE_KERNEL=1 python3 tools/r06_leg_probe.py $d 2>&1 | tail -1 | tee -a gpurun_out/r06_leg_probe.txt
SVX_BAM_DEVICE_INFLATE=0 python3 tools/r06_leg_probe.py $d 2>&1 | tail -1 | tee -a gpurun_out/r06_leg_probe.txt
