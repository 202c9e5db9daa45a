cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04b; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_cigar.py tests/test_gpu_collect.py -x -q 2>&1 | tail -5 > $o/pytest.txt
timeout 300 python tools/dense_probe.py > $o/dense_new.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/k -o s -- python3 tools/dense_probe.py --only product,knot --reps 20 > $o/prof.log 2>&1
find $o/k -name "s_kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
rm -rf $o/k
cat $o/pytest.txt; cut -c1-120 $o/kernel_stats.csv
