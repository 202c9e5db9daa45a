#!/bin/bash
# round 6, after the wave-per-member parse: the device-leg tests, then on one full-size dataset the cohort command at a few
# settings and the one-shot command's timeline at device shares 50 / 100 / 0 (and the former kernel at 50 for the difference)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_device_pool.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4
SVX_KEEP_DATASET=1 python3 tools/r06_cohort_ab.py --n 8 --out gpurun_out/r06_wave_cohort.jsonl --settings "${SETTINGS:-4:100:400:1,4:100:0:1,4:50:0:1,3:100:400:1,6:100:400:1}" > gpurun_out/r06_wave_cohort.log 2> gpurun_out/r06_wave_cohort.err
d=$(grep DATASET gpurun_out/r06_wave_cohort.log | awk '{print $2}')
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_wave_cohort.jsonl"):
    r=json.loads(l); print("workers %s group %s share %3d wait %3d: %.2f samples/s  wall %.2f s  cpu/sample %.2f  ok %s" % (r["workers"], r["group"], r["device_inflate_percent"], r["lane_wait_ms"], r["samples_per_s"], r["wall_s"], r["cpu_seconds_per_sample"], all(x is not False for x in r["vcf_matches_real_reference_digest"])))
PY
for env in "" "SVX_BAM_DEVICE_INFLATE=100" "SVX_BAM_DEVICE_INFLATE=0" "SVX_INFLATE_KERNEL=1" ""; do
  python3 tools/cli_timeline.py $d 7 $env 2>&1 | head -1
done | tee gpurun_out/r06_wave_cli.txt
