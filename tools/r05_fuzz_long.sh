#!/bin/bash
# round 5: a long differential fuzz campaign with what is left of the round's GPU budget
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_fuzz; mkdir -p $out
S=${FUZZ_S:-900}
timeout $((S+120)) python3 tools/fuzz_cigar.py --seconds $S --seed 7100000 > $out/fuzz_cigar.txt 2>&1; tail -1 $out/fuzz_cigar.txt
timeout $((S+120)) python3 tools/fuzz_other.py --seconds $S --seed 7200000 > $out/fuzz_other.txt 2>&1; tail -1 $out/fuzz_other.txt
timeout $((S+120)) python3 tools/fuzz_other.py --only collect --seconds $S --seed 7300000 > $out/fuzz_collect.txt 2>&1; tail -1 $out/fuzz_collect.txt
timeout $((S/2+120)) python3 tools/fuzz_pipeline.py --seconds $((S/2)) --seed 7400000 > $out/fuzz_pipeline.txt 2>&1; tail -1 $out/fuzz_pipeline.txt
timeout $((S/2+120)) python3 tools/fuzz_other.py --only edit --seconds $((S/2)) --seed 7500000 > $out/fuzz_edit.txt 2>&1; tail -1 $out/fuzz_edit.txt
