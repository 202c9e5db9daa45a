#!/bin/bash
# round 5, the reader's upload beside the walk (svx_bam_device_pool) and the one-round tiles' queue of 96:
# tests, the full-size run, the probe of the upload's tail, the command's HIP trace, the operating points
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_upload; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_device_pool.py tests/test_gpu_collect.py tests/test_gpu_cigar.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
d=/tmp/svx_cli_dataset
timeout 900 python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks 1 --repeat 5 > $out/e2e.json 2> $out/e2e.err
python3 -c "
import json; r=json.loads(open('$out/e2e.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('e2e: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest')))
print('   collect stages', {k2: round(v*1e3,2) for k2,v in m.get('collect_stages_s',{}).items() if not k2.endswith('cpu_s')})
print('   cli', r.get('cli_ranks'))" 2>&1 | cut -c1-600
timeout 600 python3 tools/upload_probe.py --dataset $d --repeat 7 > $out/upload_probe.json 2> $out/probe.err; cat $out/upload_probe.json; tail -3 $out/probe.err
export SVX_ORDERLY_EXIT=1
timeout 600 rocprofv3 --hip-trace --memory-copy-trace --output-format csv -d $out/hip -o s -- python3 bin/svim-asm diploid $d/wd_prof_hip $d/hap1.bam $d/hap2.bam $d/ref.fa > $out/cli_hip.log 2>&1
ls $out/hip/* | head; 
for f in $(find $out/hip -name "s_memory_copy_trace.csv"); do cp $f $out/memory_copy_trace.csv; done
for f in $(find $out/hip -name "s_hip_api_trace.csv"); do head -1 $f > $out/hip_api_trace_selected.csv; grep -i "hipMemcpyAsync\|hipEventRecord\|hipStreamWaitEvent\|hipStreamSynchronize\|hipEventSynchronize\|hipLaunchKernel\|hipModuleLaunchKernel\|hipMalloc\|hipHostMalloc" $f >> $out/hip_api_trace_selected.csv; done
rm -rf $out/hip
head -3 $out/memory_copy_trace.csv; wc -l $out/memory_copy_trace.csv $out/hip_api_trace_selected.csv; head -3 $out/hip_api_trace_selected.csv
unset SVX_ORDERLY_EXIT
for i in 1 2; do timeout 300 python3 tools/collect_probe.py latency_case product_point 2>/dev/null | tee -a $out/points.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items(): print('  ', k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in v if x in ('frac','ms_per_step','host_call_ms','host_call_pools_in_hbm_ms')})"; done
timeout 300 python3 tools/dense_probe.py 2>&1 | grep '"small"' | cut -c1-330
bash tools/r05_ab.sh default
