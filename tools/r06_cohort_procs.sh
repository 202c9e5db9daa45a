#!/bin/bash
# round 6: svim-asm-cohort as P processes on one device (each with its own workers) against one process, N=16 samples
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 tools/r06_cohort_ab.py --dataset /tmp/svx_cohort_ds --n ${N:-16} --out gpurun_out/r06_cohort_procs.jsonl --settings "${SETTINGS:-4:100:400:1:1,3:100:400:1:2,2:100:400:1:2,2:100:400:1:4,6:100:400:1:1,4:100:400:1:2,4:100:400:1:1}" > gpurun_out/r06_cohort_procs.log 2> gpurun_out/r06_cohort_procs.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_cohort_procs.jsonl"):
    r=json.loads(l); print("procs %s workers %s: %.2f samples/s  wall %.2f s  cpu/sample %.2f  rss %.0f MB  ok %s" % (r["processes"], r["workers"], r["samples_per_s"], r["wall_s"], r["cpu_seconds_per_sample"], r["peak_rss_mb_of_any_child_so_far"], all(x is not False for x in r["vcf_matches_real_reference_digest"])))
PY
