#!/bin/bash
# config 5 end to end with the PAIR distance jobs in 1 / 2 / 4 / 8 chunks; the config-5 golden; the
# full-size sample in process (VCF written by the formatting threads)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s4; mkdir -p $out
timeout 1200 python3 tools/e2e_bench.py --config5 --keep $out/c5data --repeat 1 --ranks "" > /dev/null 2> $out/gen.err
for i in 1 2; do
for k in 2 3 4; do
  SVX_PAIR_CHUNKS=$k timeout 600 python3 tools/e2e_bench.py --config5 --dataset $out/c5data --repeat 5 --ranks "" > $out/c5_chunks${k}_$i.json 2> $out/c5_chunks${k}_$i.err
  python3 -c "
import json; r=json.loads(open('$out/c5_chunks${k}_$i.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('chunks $k run $i: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest')))
print('   pair stages', {k2[5:-2]: round(v*1e3,1) for k2,v in m.get('pair_stages_s',{}).items() if not k2.endswith('cpu_s')})
print('   vcf stages', {k2[4:-2]: round(v*1e3,1) for k2,v in m.get('vcf_stages_s',{}).items() if not k2.endswith('cpu_s')})"
done
done
rm -rf $out/c5data
