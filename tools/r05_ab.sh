#!/bin/bash
# per-kernel times of the cohort step for several builds: tools/r05_ab.sh [bench args --] name...
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
extra=""
if [ "$1" = "--args" ]; then extra="$2"; shift 2; fi
for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  echo "== $v $extra"
  bash tools/kstats.sh $lib --no-extras $extra 2>&1 | grep -E "finish_a3|cigar_tiles|desc_scan|tile_alo|cigar_dense" | grep -v "calls   20"
  python3 -c "
import json; r=json.load(open('gpurun_out/ks_$(basename $lib .so)/bench.json')); rf=r['roofline']
print('   ms/step %.4f sustained %.4f kernel_ms %.4f path_ms %.4f path_frac %.3f' % (r['ms_per_step'], r.get('sustained',{}).get('ms_per_step',0), rf['kernel_ms'], rf['path_ms'], rf['path_frac']))" 2>/dev/null
done
