#!/bin/bash
# k_segments duration at cohort scale for several library builds: tools/segstats.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  out=gpurun_out/seg_$(basename $lib .so); rm -rf $out; mkdir -p $out
  SVX_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 bench.py --no-cpu-baseline --no-extras --steps 30 > /dev/null 2> $out/err.txt
  echo "== $lib"; python3 - $out <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_segments" in r["Name"] or "k_cigar_tiles" in r["Name"]:
        print("%-40s calls %5s avg %9.1f us  min %9.1f" % (r["Name"].split("k_",1)[1][:38], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
