#!/bin/bash
# round 6: svim-asm-cohort N samples, 4 workers: inflate lanes x how long a call waits for one (SVX_COHORT_LANES, SVX_COHORT_INFLATE_WAIT_MS)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for s in ${COMBOS:-2:400 2:3000 3:3000 8:400}; do
  lanes=${s%%:*}; wait=${s##*:}
  SVX_COHORT_LANES=$lanes python3 tools/r06_cohort_ab.py --dataset /tmp/svx_cohort_ds --n ${N:-24} --out gpurun_out/r06_cohort_wait_tmp.jsonl --settings "4:100:$wait:1" > /dev/null 2>> gpurun_out/r06_cohort_wait.err
  python3 -c "
import json
for l in open('gpurun_out/r06_cohort_wait_tmp.jsonl'):
    r=json.loads(l); print('lanes $lanes wait $wait ms: %.2f samples/s  wall %.2f s  cpu/sample %.2f  ok %s' % (r['samples_per_s'], r['wall_s'], r['cpu_seconds_per_sample'], all(x is not False for x in r['vcf_matches_real_reference_digest'])))"
done; done | tee gpurun_out/r06_cohort_wait.txt
