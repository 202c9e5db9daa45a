#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/svx_infl_ds
python3 -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -3
for k in ls lane ls lane; do
  echo "== kernel $k"
  SVX_INFLATE_KERNEL=$k python3 tools/gpu_inflate_probe.py --scale 0.25 --dataset /tmp/svx_infl_ds --members 16000 --min-payload 8192 --counts 1000,3000,7261,14000 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
done
