#!/bin/bash
# round 6, after the wave-per-member parse: the one-shot command on the full-size sample, 11 fresh processes per device share,
# interleaved twice; then the whole timeline at the default
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=/tmp/svx_e2e_ds; rm -rf $d
python3 tools/e2e_bench.py --scale 1.0 --keep $d --ranks "" --repeat 1 > /dev/null 2> gpurun_out/r06_wave_cli.err
for rep in 1 2; do for share in ${SHARES:-50 75 100 0}; do
  python3 tools/cli_timeline.py $d 11 SVX_BAM_DEVICE_INFLATE=$share 2>&1 | grep -E "^wall-clock|^CPU seconds" | tr '\n' ' '; echo
done; done | tee gpurun_out/r06_wave_cli_shares.txt
python3 tools/cli_timeline.py $d 11 > gpurun_out/r06_wave_cli_timeline.txt 2>&1; head -16 gpurun_out/r06_wave_cli_timeline.txt
