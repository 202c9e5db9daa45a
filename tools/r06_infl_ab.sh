#!/bin/bash
# tools/r06_infl_ab.sh [kernel...] : the 7 261-member SEQ probe with the device decoder's forms: 3 wave-per-member parse (shipped),
# 2 lane-per-member parse, 1 one launch.  SCALE=1.0 COUNTS=7261,14000,28000 for the full-size sample's members.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/svx_infl_ds${SCALE:-}
for k in "$@"; do
  echo "== SVX_INFLATE_KERNEL=$k"
  SVX_INFLATE_KERNEL=$k python3 tools/gpu_inflate_probe.py --scale ${SCALE:-0.25} --dataset /tmp/svx_infl_ds${SCALE:-} --members ${MEMBERS:-16000} --min-payload 8192 --counts ${COUNTS:-1000,3000,7261} 2>&1 | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
done
