#!/usr/bin/env python3
"""Per-worker phase timeline of one `svim-asm-cohort diploid` process over N copies of a sample (GPU box):
    python tools/cohort_timeline.py DATASET_DIR N [--cohort_workers K ...]
Prints for every worker the phases of its groups (wall offsets from the process start), how much of the run's wall-clock
each phase kind takes per sample, and the CPU seconds of the whole process between consecutive marks (all threads)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, n, extra = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    out = tempfile.mkdtemp(prefix="svx_ct_")
    manifest = os.path.join(out, "m.txt")
    with open(manifest, "w") as f:
        for k in range(n):
            c = os.path.join(out, "c%d" % k)
            os.makedirs(c)
            for name in ("hap1.bam", "hap2.bam", "hap1.bam.bai", "hap2.bam.bai"):
                shutil.copyfile(os.path.join(d, name), os.path.join(c, name))
            f.write("%s %s %s\n" % (os.path.join(c, "wd"), os.path.join(c, "hap1.bam"), os.path.join(c, "hap2.bam")))
    tl = os.path.join(out, "tl.jsonl")
    env = dict(os.environ, SVX_CLI_TIMELINE=tl)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "svim-asm-cohort"), "diploid", manifest, os.path.join(d, "ref.fa")] + extra,
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    wall = time.time() - t0
    marks = [json.loads(l) for l in open(tl)]
    print("rc %d wall %.3f s  %.2f samples/s  process CPU-s %.2f" % (p.returncode, wall, n / wall, marks[-1]["cpu"]))
    by_thread = {}
    for m in marks:
        by_thread.setdefault(m["thread"], []).append(m)
    kinds = {}
    for th, ms in sorted(by_thread.items()):
        line = []
        prev = None
        for m in ms:
            line.append("%s@%.3f" % (m["name"].split()[0] + (str(m.get("sample", "")) if "sample" in m else ""), m["t"] - t0))
            if prev is not None and th.startswith("cohort-"):
                kinds.setdefault(m["name"], []).append(m["t"] - prev["t"])
            prev = m
        print(th, " ".join(line))
    for k, v in kinds.items():
        print("  %-14s mean %.3f s  max %.3f  (n=%d)" % (k, sum(v) / len(v), max(v), len(v)))
    shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main()
