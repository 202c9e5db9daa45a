#!/bin/bash
# round 5: k_bgzf_inflate, members per workgroup / per wave / waves per SIMD, on the SEQ members of a half-scale haplotype
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_infl; mkdir -p $out
d=/tmp/svx_infl_ds5; mkdir -p $d
for v in default infl_16_2_w8 infl_16_2_w6 infl_8_1_w8 infl_16_4_w6 infl_32_4_w8; do
  lib=svim_asm_amd/libsvx.so; [ $v != default ] && lib=build/libsvx_$v.so
  SVX_LIB=$PWD/$lib timeout 600 python3 tools/gpu_inflate_probe.py --scale 0.5 --dataset $d --members 14000 --min-payload 8192 --counts 3400,6800,13600 > $out/geom2_$v.json 2> $out/err.txt
  python3 -c "
import json; r=json.load(open('$out/geom2_$v.json')); print('$v', r['kernel_ms_by_member_count'], 'all', r['members'], round(r['device_kernel_ms'],2), 'ok', r['all_status_ok_and_bytes_equal_zlib_on_sample'])" || tail -3 $out/err.txt
done
