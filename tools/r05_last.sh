#!/bin/bash
# round 5, last call: config 5 end to end on the final tree, then fuzz campaigns with what is left
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_last; mkdir -p $out
timeout 1200 python3 tools/e2e_bench.py --config5 --repeat 5 --ranks "1" > $out/e2e_config5.json 2> $out/c5.err
python3 -c "
import json; r=json.loads(open('$out/e2e_config5.json').read().strip().splitlines()[-1]); m=r.get('median_run', r)
print('config5: total %.4f all %s collect %.4f pair %.4f vcf %.4f ok %s cli %s' % (m['product_total_s'], [round(x,3) for x in r.get('all_runs_total_s',[])], m['collect_s'], m['pair_s'], m['vcf_s'], r.get('vcf_matches_real_reference_digest'), r.get('cli_all_wall_s')))"
S=${FUZZ_S:-600}
timeout $((S+120)) python3 tools/fuzz_other.py --only collect --seconds $S --seed 6100000 > $out/fuzz_collect.txt 2>&1; tail -1 $out/fuzz_collect.txt
timeout $((S+120)) python3 tools/fuzz_cigar.py --seconds $S --seed 6200000 > $out/fuzz_cigar.txt 2>&1; tail -1 $out/fuzz_cigar.txt
timeout $((S+120)) python3 tools/fuzz_other.py --seconds $S --seed 6300000 > $out/fuzz_other.txt 2>&1; tail -1 $out/fuzz_other.txt
timeout $((S/2+120)) python3 tools/fuzz_pipeline.py --seconds $((S/2)) --seed 6400000 > $out/fuzz_pipeline.txt 2>&1; tail -1 $out/fuzz_pipeline.txt
