#!/bin/bash
# round 6, one dataset generation: svim-asm-cohort settings A/B (tools/r06_cohort_ab.py), then the per-phase timeline of the
# one-shot command on the same files (tools/cli_timeline.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SVX_KEEP_DATASET=1 python3 tools/r06_cohort_ab.py --n 8 --out gpurun_out/r06_cohort_ab.jsonl "$@" > gpurun_out/r06_cohort_ab.log 2> gpurun_out/r06_cohort_ab.err
d=$(grep DATASET gpurun_out/r06_cohort_ab.log | awk '{print $2}')
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_cohort_ab.jsonl"):
    r=json.loads(l); print("workers %s group %s share %3d wait %3d: %.2f samples/s  wall %.2f s  cpu/sample %.2f  rss %.0f MB  ok %s" % (r["workers"], r["group"], r["device_inflate_percent"], r["lane_wait_ms"], r["samples_per_s"], r["wall_s"], r["cpu_seconds_per_sample"], r["peak_rss_mb_of_any_child_so_far"], all(x is not False for x in r["vcf_matches_real_reference_digest"]) and (r["rc"]==0 or r["rc"]==[0]*len(r["rc"]) if isinstance(r["rc"], list) else r["rc"]==0)))
PY
python3 tools/cli_timeline.py $d 5 > gpurun_out/r06_cli_timeline.txt 2>&1
cat gpurun_out/r06_cli_timeline.txt
