#!/bin/bash
# per-kernel average durations of one bench run under rocprofv3: tools/kstats.sh lib.so [bench args]
lib=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ks_$(basename $lib .so); rm -rf $out; mkdir -p $out
SVX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.txt
python3 - $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/s_kernel_stats.csv")):
    n = r["Name"]
    if "k_" in n and "at::" not in n:
        print("%-28s calls %4s avg %9.1f us  min %9.1f" % (n.split("k_")[1].split("(")[0].split("<")[0][:26], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
