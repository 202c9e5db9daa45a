cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04k; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_cigar.py tests/test_gpu_collect.py tests/test_gpu_ctx.py tests/test_gpu_segments.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -15 > $o/pytest.txt
timeout 900 python bench.py --no-cpu-baseline --e2e-scale 0 > $o/bench.json 2> $o/bench.err
for leg in latency_case product_point; do
rocprofv3 --kernel-trace --stats --output-format csv -d $o/k_$leg -o s -- python3 tools/collect_probe.py $leg > $o/$leg.log 2>&1
find $o/k_$leg -name "s_kernel_stats.csv" -exec cp {} $o/${leg}_kernel_stats.csv \;
rm -rf $o/k_$leg
done
cat $o/pytest.txt; tail -5 $o/bench.err
cut -d, -f1-4,6-8 $o/latency_case_kernel_stats.csv | cut -c1-200; cut -d, -f1-4,6-8 $o/product_point_kernel_stats.csv | cut -c1-200
