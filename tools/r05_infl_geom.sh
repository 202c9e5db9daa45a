#!/bin/bash
# round 5: k_bgzf_inflate with fewer members per wave (SVX_INFL_LANES = members per workgroup, SVX_INFL_ACTIVE = per wave)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_infl; mkdir -p $out
d=/tmp/svx_infl_ds; mkdir -p $d
for v in default infl_16_4 infl_8_2 infl_16_2 infl_4_1 infl_8_1; do
  lib=svim_asm_amd/libsvx.so; [ $v != default ] && lib=build/libsvx_$v.so
  SVX_LIB=$PWD/$lib timeout 600 python3 tools/gpu_inflate_probe.py --scale 0.25 --dataset $d --members 14000 --min-payload 8192 --counts 1000,3000,7261 > $out/geom_$v.json 2> $out/err.txt
  python3 -c "
import json; r=json.load(open('$out/geom_$v.json')); print('$v', r['kernel_ms_by_member_count'], 'all', round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])" || tail -3 $out/err.txt
done
