#!/bin/bash
# tools/r06_infl_ab.sh name... : the 7 261-member SEQ probe on several builds (build/libsvx_<name>.so; `default` = the tree's)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/svx_infl_ds
for v in "$@"; do lib=build/libsvx_$v.so; [ $v = default ] && lib=svim_asm_amd/libsvx.so
  echo "== $v"
  SVX_LIB=$PWD/$lib python3 tools/gpu_inflate_probe.py --scale 0.25 --dataset /tmp/svx_infl_ds --members 16000 --min-payload 8192 --counts 1000,3000,7261 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
done
