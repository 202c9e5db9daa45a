#!/bin/bash
# round 5, step 1: parity of the new chain rows, then per-kernel times of the cohort step for several builds
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s1; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_collect.py tests/test_gpu_cigar.py tests/test_gpu_segments.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  for deal in table equal; do
    [ "$v" = r04base ] && [ "$deal" = table ] && continue
    echo "== $v deal=$deal"
    bash tools/kstats.sh $lib --no-extras --chain-deal $deal 2>&1 | grep -E "finish_a3|cigar_tiles|desc_scan|tile_alo|cigar_dense"
    python3 -c "
import json; r=json.load(open('gpurun_out/ks_$(basename $lib .so)/bench.json')); rf=r['roofline']
print('   ms/step %.4f sustained %s kernel_ms %.4f path_ms %.4f path_frac %.3f' % (r['ms_per_step'], r.get('sustained',{}).get('ms_per_step'), rf['kernel_ms'], rf['path_ms'], rf['path_frac']))"
  done
done
