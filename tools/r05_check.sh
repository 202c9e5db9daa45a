#!/bin/bash
# round 5: the goldens that run PAIR (config 5, medium, large, full) and config 5's wall-clock after a change to PAIR's host side
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_check; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_config5_golden.py tests/test_medium_golden.py tests/test_large_golden.py tests/test_full_golden.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
timeout 900 python3 tools/e2e_bench.py --config5 --repeat 7 --ranks "" > $out/e2e_config5.json 2> $out/c5.err
python3 -c "
import json; r=json.loads(open('$out/e2e_config5.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('config5: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest')))
print('   pair stages', {k2[5:-2]: round(v*1e3,1) for k2,v in m.get('pair_stages_s',{}).items() if not k2.endswith('cpu_s')})
print('   collect stages', {k2: round(v*1e3,1) for k2,v in m.get('collect_stages_s',{}).items() if not k2.endswith('cpu_s')})"
