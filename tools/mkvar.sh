#!/bin/bash
# build an ablation / variant library: tools/mkvar.sh name -DFLAG [-DFLAG...]  -> build/libsvx_<name>.so
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build
python3 - "$name" "$@" <<'PY'
import sys
sys.path.insert(0, ".")
from svim_asm_amd import build
print(build.build_lib(out="build/libsvx_%s.so" % sys.argv[1], defines=sys.argv[2:]))
PY
