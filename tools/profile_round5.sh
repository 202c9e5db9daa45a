#!/bin/bash
# Round-5 evidence on the GPU box.  Output under gpurun_out/$1/ (default r05); tools/sync_profiles.py (or a plain cp)
# moves what is judged into profiles/r05_*.
#   part A: bench.json (default line, all legs), kernel_stats.csv (rocprofv3 --kernel-trace --stats of the headline
#           command, no extras), points_*_kernel_stats.csv (the two product operating points), dense_* (SV-dense batches)
#   part B (with "pmc" as $2): counter passes, separate runs: pmc/ (cohort step), pmc_points/ (product operating
#           points), pmc_dense/ (a batch whose dense-tile launch is not empty)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r05}; mkdir -p $out
if [ "$2" != "pmc" ]; then
timeout 1800 python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
find $out/stats -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
for leg in latency_case product_point; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/k_$leg -o s -- python3 tools/collect_probe.py $leg > $out/$leg.json 2>> $out/rocprof.err
  find $out/k_$leg -name "s_kernel_stats.csv" -exec cp {} $out/points_${leg}_kernel_stats.csv \;
  rm -rf $out/k_$leg
done
timeout 600 python3 tools/dense_probe.py > $out/dense_probe.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k_dense -o s -- python3 tools/dense_probe.py --only product,knot --reps 20 > /dev/null 2>> $out/rocprof.err
find $out/k_dense -name "s_kernel_stats.csv" -exec cp {} $out/dense_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k_ed -o s -- python3 tools/kbench.py pair editdist > $out/kbench_under_rocprof.json 2>> $out/rocprof.err
find $out/k_ed -name "s_kernel_stats.csv" -exec cp {} $out/kernel_stats_pair_editdist.csv \;
timeout 900 python3 tools/gpu_inflate_probe.py --scale 0.25 --members 20000 > $out/gpu_inflate_probe.json 2>> $out/rocprof.err
rm -rf $out/stats $out/k_dense $out/k_ed
tail -c 400 $out/bench.json
else
bash tools/pmc.sh $out/pmc > $out/pmc.txt 2>&1
PMC_CMD="tools/collect_probe.py latency_case product_point" bash tools/pmc.sh $out/pmc_points > $out/pmc_points.txt 2>&1
PMC_CMD="tools/dense_probe.py --only product,knot --reps 10" bash tools/pmc.sh $out/pmc_dense > $out/pmc_dense.txt 2>&1
rm -f $out/pmc*/pass*_kernel_trace.csv $out/pmc*/pass*_agent_info.csv
tail -5 $out/pmc.txt | cut -c1-300
fi
