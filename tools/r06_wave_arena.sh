#!/bin/bash
# A/B: members per slice of launches (the token arena's size) — kernel time by member count, variants arena4k / default (6144) / arena16k / arena32k
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  echo -n "$v: "; SVX_LIB=$PWD/$lib SCALE=1.0 MEMBERS=40000 COUNTS=3000,7261,14000,28000 bash tools/r06_infl_ab.sh 3 | tail -1
done; done
