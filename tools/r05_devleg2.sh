#!/bin/bash
# round 5: the device leg as the default (quota-aware share): whole -m gpu suite, the command line with and without it
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_devleg; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=5 > $out/pytest_gpu.txt 2>&1; tail -9 $out/pytest_gpu.txt
d=/tmp/svx_cli_dataset
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
for i in 1 2; do for v in default 0; do
  if [ $v = default ]; then unset SVX_BAM_DEVICE_INFLATE; else export SVX_BAM_DEVICE_INFLATE=0; fi
  echo "== command line, SVX_BAM_DEVICE_INFLATE=$v"; python3 tools/cli_timeline.py $d 5
done; done
unset SVX_BAM_DEVICE_INFLATE
SVX_BAM_DEBUG=1 python3 bin/svim-asm diploid /tmp/wd_dbg $d/hap1.bam $d/hap2.bam $d/ref.fa 2>&1 | grep "device leg\|slices, " | cut -c1-300
