#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=$(python3 -c "
import tempfile,sys
sys.path.insert(0,'.')
from svim_asm_amd import synth_bam
from tools import e2e_bench
d=tempfile.mkdtemp(prefix='svx_ds_'); synth_bam.write_dataset(d, **e2e_bench.dataset_args(1.0)); print(d)" 2>/dev/null | tail -1)
for kt in "3 10" "3 3" "3 4" "4 2" "4 3" "6 2" "8 2" "8 1" "4 4" "6 3"; do set -- $kt
  echo "== workers $1 threads $2"; python3 tools/cohort_timeline.py $d 8 --cohort_workers $1 --cohort_threads $2 | head -1
done
for kt in "4 3" "6 2" "3 4"; do set -- $kt
  echo "== N=16 workers $1 threads $2"; python3 tools/cohort_timeline.py $d 16 --cohort_workers $1 --cohort_threads $2 | head -1
done
# how long does a process take to leave with the BAMs mapped and touched?
python3 - $d <<'PY'
import mmap, os, subprocess, sys, time
d = sys.argv[1]
code = r'''
import mmap, os, sys, time
ms = []
for name in ("hap1.bam", "hap2.bam"):
    f = open(os.path.join(sys.argv[1], name), "rb")
    m = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ)
    s = 0
    for off in range(0, len(m), 4096):
        s += m[off]
    ms.append(m)
if sys.argv[2] == "unmap":
    t = time.time()
    for m in ms: m.close()
    sys.stderr.write("unmap %.3f\n" % (time.time() - t))
sys.stderr.write("EXIT %.6f\n" % time.time())
os._exit(0)
'''
for mode in ("keep", "unmap", "keep"):
    p = subprocess.run([sys.executable, "-c", code, d, mode], stderr=subprocess.PIPE, text=True)
    t1 = time.time()
    t_exit = [float(l.split()[1]) for l in p.stderr.split("\n") if l.startswith("EXIT")][0]
    print("mode %s: process gone %.3f s after os._exit  %s" % (mode, t1 - t_exit, [l for l in p.stderr.split("\n") if l.startswith("unmap")]))
PY
