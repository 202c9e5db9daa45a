#!/bin/bash
# tools/sp_ab.sh [sp_check args --] name...   : tools/sp_check.py on several builds (build/libsvx_<name>.so; `default` = the tree's)
extra=""
if [ "$1" = "--args" ]; then extra="$2"; shift 2; fi
for v in "$@"; do
  lib=build/libsvx_$v.so; [ "$v" = default ] && lib=svim_asm_amd/libsvx.so
  echo "== $v"
  SVX_LIB=$PWD/$lib python tools/sp_check.py --reps 1 $extra 2>&1 | tail -1
done
