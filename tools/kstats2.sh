#!/bin/bash
# per-kernel durations of tools/kbench.py legs under rocprofv3: tools/kstats2.sh <outdir> leg [leg...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/kbench.py "$@" > $out/kbench.json 2> $out/err.txt
python3 - $out <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "k_" in n and "at::" not in n:
        print("%-40s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n.split("k_", 1)[1].split("(")[0][:38], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
