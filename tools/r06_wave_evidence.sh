#!/bin/bash
# round 6, the shipped device decoder (wave-per-member parse): everything the evidence under profiles/r06_inflate_wave_* comes
# from, on one box.  Output: gpurun_out/r06_wave/*
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_wave; mkdir -p $out
# 1. rocprofv3 kernel-trace stats + two PMC passes on the 7 261-member probe, the three forms
for k in 3 2; do timeout 600 bash tools/r06_inflate_evidence.sh $k > $out/evidence_$k.log 2>&1; done
cp gpurun_out/r06_infl/summary.json $out/shipped_summary.json; cp gpurun_out/r06_infl_2/summary.json $out/lane_parse_summary.json
find gpurun_out/r06_infl/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# 2. per-phase clocks of the parse (a -DSVX_WPARSE_STATS build), the probe's members as written (level 1) and recompressed at 6
for lv in "" "--level 6"; do SVX_LIB=$PWD/build/libsvx_wstats.so python3 tools/r06_wave_stats.py --dataset /tmp/svx_infl_ds $lv 2>&1 | tail -1; done > $out/phases.jsonl
# 3. every member of both haplotype BAMs of the full-size sample against zlib, and the forms by member count
for hap in 1 2; do python3 tools/gpu_inflate_probe.py --scale 1.0 --dataset /tmp/svx_infl_ds1.0 --members 1000000 --hap $hap --check-all 2>&1 | tail -1; done > $out/all_members.jsonl
SCALE=1.0 MEMBERS=40000 COUNTS=1000,3000,7261,14000,28000 bash tools/r06_infl_ab.sh 3 2 1 > $out/forms_by_count.txt 2>&1
# 4. fuzzers
timeout 400 python3 tools/fuzz_other.py --only inflate --seconds 150 --seed 61 > $out/fuzz_inflate.txt 2>&1
timeout 400 python3 tools/fuzz_bam_reader.py --seconds 120 --seed 62 > $out/fuzz_bam_reader.txt 2>&1
tail -2 $out/forms_by_count.txt $out/fuzz_inflate.txt $out/fuzz_bam_reader.txt; cat $out/phases.jsonl | cut -c1-400; cut -c1-600 $out/all_members.jsonl
