#!/bin/bash
# usage: tools/pmc_lib.sh lib.so  -> VALU/SALU/LDS/wave-cycle counters of k_cigar_tiles
lib=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcl_$(basename $lib .so); mkdir -p $out
SVX_LIB=$PWD/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - $out <<'PY'
import csv, collections, sys
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1]+"/p_counter_collection.csv")):
    if "k_cigar_tiles" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); agg["dur"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
print({k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
