#!/bin/bash
# A/B of variant builds of the wave-per-member parse (tools/mkvar.sh NAME -D...): tools/r06_wave_ab.sh NAME...  ("default": the tree's library)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  echo -n "$v: "; SVX_LIB=$PWD/$lib bash tools/r06_infl_ab.sh 3 | tail -1
done; done
