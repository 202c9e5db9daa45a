#!/bin/bash
# round 6: svim-asm-cohort, inflate lanes on the device (SVX_COHORT_LANES) x workers, N samples, one process
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lanes in ${LANES:-2 4 8}; do
  SVX_COHORT_LANES=$lanes python3 tools/r06_cohort_ab.py --dataset /tmp/svx_cohort_ds --n ${N:-24} --out gpurun_out/r06_cohort_lanes_$lanes.jsonl --settings "${SETTINGS:-4:100:400:1,6:100:400:1,8:100:400:1}" > /dev/null 2>> gpurun_out/r06_cohort_lanes.err
  python3 -c "
import json
for l in open('gpurun_out/r06_cohort_lanes_$lanes.jsonl'):
    r=json.loads(l); print('lanes $lanes workers %s: %.2f samples/s  wall %.2f s  cpu/sample %.2f  rss %.0f MB  ok %s' % (r['workers'], r['samples_per_s'], r['wall_s'], r['cpu_seconds_per_sample'], r['peak_rss_mb_of_any_child_so_far'], all(x is not False for x in r['vcf_matches_real_reference_digest'])))"
done | tee gpurun_out/r06_cohort_lanes.txt
