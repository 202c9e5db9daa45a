#!/bin/bash
# A/B of a decoder variant (SVX_LIB) against the tree's on the SEQ members of files written by zlib level 1 / 6 and libdeflate 6
cd "$GRAFT_REPO_ROOT"
for lv in 1 106 6; do
  d=/tmp/svx_ds_lv$lv
  [ -f $d/hap1.bam ] || python3 tools/e2e_bench.py --scale 0.25 --bam-level $lv --keep $d --ranks "" --no-in-process > /dev/null 2>&1
  for v in "$@"; do
    lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
    echo -n "level $lv $v: "; SVX_LIB=$PWD/$lib python3 tools/gpu_inflate_probe.py --scale 0.25 --dataset $d --members 16000 --min-payload 8192 --counts 1000,7000 2>/dev/null | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])"
  done
done
