#!/bin/bash
# round 5, step 6: full-size in-process runs after the FASTA unmapping moved behind the VCF write (+ the command line)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s6; mkdir -p $out
timeout 1500 python3 tools/e2e_bench.py --scale 1.0 --repeat 7 --ranks "1" > $out/full.json 2> $out/full.err
python3 -c "
import json; r=json.loads(open('$out/full.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('full: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s optout %s cli %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest'), r.get('prefix_only_no_crc_total_s'), r.get('cli_all_wall_s')))
print('   vcf stages', {k2[4:-2]: round(v*1e3,1) for k2,v in m.get('vcf_stages_s',{}).items() if not k2.endswith('cpu_s')})
print('   collect stages', {k2: round(v*1e3,1) for k2,v in m.get('collect_stages_s',{}).items()})
print('   cpu', m.get('cpu_seconds'), 'optout cpu', r.get('prefix_only_no_crc_cpu_seconds'))"
