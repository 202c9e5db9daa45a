#!/bin/bash
# round 5 measurement hygiene (verdict item 7): --distinct 8 vs 256 interleaved, a 1-rank nccl (RCCL) process group,
# PMC passes of the pair sort and of the edit-distance kernels
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_hyg; mkdir -p $out
for i in 1 2; do
  for d in 8 256; do
    timeout 900 python3 bench.py --no-extras --no-cpu-baseline --distinct $d > $out/bench_distinct${d}_$i.json 2> $out/err_distinct${d}_$i.txt
    python3 -c "
import json; r=json.loads(open('$out/bench_distinct${d}_$i.json').read().strip().splitlines()[-1]); rf=r['roofline']
print('distinct $d run $i: ms/step %.4f median-of-5 value %.4g sustained %.4f kernel_ms %.4f frac %.3f path_frac %.3f step_frac %.3f' % (r['ms_per_step'], r['value_median_of_5'], r['sustained']['ms_per_step'], rf['kernel_ms'], rf['frac'], rf['path_frac'], rf['step_frac']))"
  done
done
timeout 600 python3 bench.py --group-at-1 --backend nccl --no-extras --no-cpu-baseline > $out/bench_nccl_group_at_1.json 2> $out/err_nccl1.txt
echo "nccl group at 1 rank rc=$?"; python3 -c "
import json; r=json.loads(open('$out/bench_nccl_group_at_1.json').read().strip().splitlines()[-1]); print({k: r[k] for k in ('value','n_gpus','backend','collective_world_verified','distinct_devices')})"
bash tools/pmc_pair.sh $out/pmc_pair > $out/pmc_pair.txt 2>&1; tail -4 $out/pmc_pair.txt | cut -c1-400
PMC_CMD="tools/kbench.py editdist" bash tools/pmc.sh $out/pmc_edit > $out/pmc_edit.txt 2>&1; tail -4 $out/pmc_edit.txt | cut -c1-400
rm -f $out/pmc*/pass*_kernel_trace.csv $out/pmc*/pass*_agent_info.csv
