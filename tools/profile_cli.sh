#!/bin/bash
# HIP-API and kernel summaries of `svim-asm diploid` on the full-size sample (GPU box):
#   $1 = output directory under gpurun_out/ (default r03_cli); copies wanted go to profiles/.
#   hip_api_stats.csv   rocprofv3 --hip-trace --stats: how many hipMemcpy* / hipStreamSynchronize the command issues
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
cd /tmp && export TMPDIR=/tmp SVX_ORDERLY_EXIT=1; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r04_cli}; mkdir -p $out
d=/tmp/svx_cli_dataset
python3 tools/e2e_bench.py --scale ${2:-1.0} --keep $d --ranks 1 --repeat 2 > $out/e2e.json 2> $out/e2e.err
rocprofv3 --hip-trace --stats --output-format csv -d $out/hip -o s -- python3 bin/svim-asm diploid $d/wd_prof_hip $d/hap1.bam $d/hap2.bam $d/ref.fa > $out/cli_hip.log 2>&1
find $out/hip -name "s_hip_api_stats.csv" -exec cp {} $out/hip_api_stats.csv \;
find $out/hip -name "s_hip_api_trace.csv" -exec sh -c 'cut -d, -f1-4 "$1" | grep -i "memcpy\|Synchronize" > '$out'/hip_memcpy_sync_calls.csv' _ {} \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o s -- python3 bin/svim-asm diploid $d/wd_prof_k $d/hap1.bam $d/hap2.bam $d/ref.fa > $out/cli_k.log 2>&1
find $out/k -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/hip $out/k
head -40 $out/hip_api_stats.csv; wc -l $out/hip_memcpy_sync_calls.csv; cat $out/kernel_stats.csv | cut -c1-150 | head -30
