cd /tmp && export TMPDIR=/tmp SVX_ORDERLY_EXIT=1; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04o; mkdir -p $o
d=/tmp/svx_c5
python3 tools/e2e_bench.py --config5 --keep $d --ranks 1 --repeat 3 > $o/e2e_c5.json 2> $o/e2e_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/k -o s -- python3 bin/svim-asm diploid $d/wd_prof_k $d/hap1.bam $d/hap2.bam $d/ref.fa > $o/cli_k.log 2>&1
find $o/k -name "s_kernel_stats.csv" -exec cp {} $o/c5_kernel_stats.csv \;
rocprofv3 --hip-trace --stats --output-format csv -d $o/h -o s -- python3 bin/svim-asm diploid $d/wd_prof_h $d/hap1.bam $d/hap2.bam $d/ref.fa > $o/cli_h.log 2>&1
find $o/h -name "s_hip_api_stats.csv" -exec cp {} $o/c5_hip_api_stats.csv \;
rm -rf $o/k $o/h
cut -d, -f1-4 $o/c5_kernel_stats.csv | cut -c1-160 | head -24; head -14 $o/c5_hip_api_stats.csv | cut -c1-150
