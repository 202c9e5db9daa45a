#!/bin/bash
# round 5, step 5: full-size in-process runs (VCF stage clocks), collect fuzz campaign on the final chain rows
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s5; mkdir -p $out
timeout 1500 python3 tools/e2e_bench.py --scale 1.0 --repeat 7 --ranks "" > $out/full.json 2> $out/full.err
python3 -c "
import json; r=json.loads(open('$out/full.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('full: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s optout %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest'), r.get('prefix_only_no_crc_total_s')))
print('   vcf stages', {k2[4:-2]: round(v*1e3,1) for k2,v in m.get('vcf_stages_s',{}).items() if not k2.endswith('cpu_s')})
print('   cpu', m.get('cpu_seconds'))"
timeout 900 python3 tools/fuzz_other.py --only collect --seconds ${FUZZ_S:-300} --seed 5100000 > $out/fuzz_collect.txt 2>&1; tail -3 $out/fuzz_collect.txt
timeout 600 python3 tools/fuzz_other.py --only edit --seconds 120 --seed 5200000 > $out/fuzz_edit.txt 2>&1; tail -2 $out/fuzz_edit.txt
timeout 600 python3 tools/fuzz_pipeline.py --seconds 120 --seed 5300000 > $out/fuzz_pipeline.txt 2>&1; tail -2 $out/fuzz_pipeline.txt
timeout 600 python3 tools/fuzz_cigar.py --seconds 120 --seed 5400000 > $out/fuzz_cigar.txt 2>&1; tail -2 $out/fuzz_cigar.txt
