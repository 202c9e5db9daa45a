#!/bin/bash
# round 5, the last tree (device leg without spinning waits): the -m gpu suite, the bench line, the driver's command, the
# command's HIP trace and kernel stats
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05d; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=5 > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 1800 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 200 $out/bench.json; echo
for i in 1 2; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_driver_cmd_$i.json 2>> $out/bench.err; python3 -c "
import json; r=json.loads(open('$out/bench_driver_cmd_$i.json').read().strip().splitlines()[-1]); rf=r['roofline']
print('driver cmd $i: value %.4g ms/step %.4f median5 %.4g kernel_ms %.4f frac %.3f path_frac %.3f step_frac %.3f throttled %s' % (r['value'], r['ms_per_step'], r['value_median_of_5'], rf['kernel_ms'], rf['frac'], rf['path_frac'], rf['step_frac'], r['host_throttled_ms_in_timed_region']))"; done
d=/tmp/svx_cli_dataset
[ -f $d/hap1.bam ] || timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2>&1
export SVX_ORDERLY_EXIT=1
timeout 600 rocprofv3 --hip-trace --memory-copy-trace --stats --output-format csv -d $out/hip -o s -- python3 bin/svim-asm diploid $d/wd_prof_hip $d/hap1.bam $d/hap2.bam $d/ref.fa > $out/cli_hip.log 2>&1
for f in $(find $out/hip -name "s_memory_copy_trace.csv"); do cp $f $out/cli_memory_copy_trace.csv; done
for f in $(find $out/hip -name "s_hip_api_stats.csv"); do cp $f $out/cli_hip_api_stats.csv; done
for f in $(find $out/hip -name "s_hip_api_trace.csv"); do head -1 $f > $out/cli_hip_api_trace_selected.csv; grep -i "hipMemcpy\|hipEventRecord\|hipStreamWaitEvent\|hipStreamSynchronize\|hipEventSynchronize\|hipLaunchKernel\|hipModuleLaunchKernel\|hipMalloc\|hipHostMalloc\|hipStreamCreate" $f >> $out/cli_hip_api_trace_selected.csv; done
rm -rf $out/hip
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o s -- python3 bin/svim-asm diploid $d/wd_prof_k $d/hap1.bam $d/hap2.bam $d/ref.fa > $out/cli_k.log 2>&1
for f in $(find $out/k -name "s_kernel_stats.csv"); do cp $f $out/cli_kernel_stats.csv; done
rm -rf $out/k
head -8 $out/cli_kernel_stats.csv | cut -c1-140
