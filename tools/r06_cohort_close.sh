#!/bin/bash
# round 6: svim-asm-cohort N = 16, readers closed on a thread of their own (default) against closed by the worker (SVX_COHORT_SYNC_CLOSE=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do for sync in "" 1; do
  SVX_COHORT_SYNC_CLOSE=$sync python3 tools/r06_cohort_ab.py --dataset /tmp/svx_cohort_ds --n ${N:-16} --out gpurun_out/r06_cohort_close_tmp.jsonl --settings "4:100:3000:1" > /dev/null 2>> gpurun_out/r06_cohort_close.err
  python3 -c "
import json
for l in open('gpurun_out/r06_cohort_close_tmp.jsonl'):
    r=json.loads(l); print('close %s: %.2f samples/s  wall %.2f s  cpu/sample %.2f  ok %s' % ('by the worker' if '$sync' else 'on a thread ', r['samples_per_s'], r['wall_s'], r['cpu_seconds_per_sample'], all(x is not False for x in r['vcf_matches_real_reference_digest'])))"
done; done | tee gpurun_out/r06_cohort_close.txt
