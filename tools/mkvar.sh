#!/bin/bash
# build an ablation / variant library: tools/mkvar.sh name -DFLAG [-DFLAG...]  -> build/libsvx_<name>.so
name=$1; shift
mkdir -p build
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -I include -I svim_asm_amd/csrc -o build/libsvx_$name.so svim_asm_amd/csrc/svx_ctx.hip svim_asm_amd/csrc/svx_cigar.hip \
  svim_asm_amd/csrc/svx_segments.hip svim_asm_amd/csrc/svx_pair.hip svim_asm_amd/csrc/svx_editdist.hip svim_asm_amd/csrc/svx_linkage.hip svim_asm_amd/csrc/svx_postpass.hip svim_asm_amd/csrc/svx_bam.cpp \
  -lz -ldl -lpthread && echo build/libsvx_$name.so
