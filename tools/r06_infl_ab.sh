#!/bin/bash
# tools/r06_infl_ab.sh [kernel...] : the 7 261-member SEQ probe with the two-pass (2) and the one-launch (1) device decoder
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/svx_infl_ds
for k in "$@"; do
  echo "== SVX_INFLATE_KERNEL=$k"
  SVX_INFLATE_KERNEL=$k python3 tools/gpu_inflate_probe.py --scale 0.25 --dataset /tmp/svx_infl_ds --members 16000 --min-payload 8192 --counts 1000,3000,7261 2>&1 | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
done
