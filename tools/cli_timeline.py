#!/usr/bin/env python3
"""Timeline of one `svim-asm diploid` as a fresh process (GPU box): process start → first log line → steps → last log
line → the moment the command calls os._exit → the moment the parent sees it gone.
    python tools/cli_timeline.py DIR_WITH_hap1.bam_hap2.bam_ref.fa [repeats]"""
import datetime
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    wd = tempfile.mkdtemp(prefix="svx_tl_")
    env = dict(os.environ, SVX_EXIT_MARK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "svim-asm"), "diploid", wd, os.path.join(d, "hap1.bam"),
                        os.path.join(d, "hap2.bam"), os.path.join(d, "ref.fa")], env=env, stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, text=True)
    t1 = time.time()
    marks = []
    for line in open(glob.glob(os.path.join(wd, "*.log"))[0]):
        m = re.match(r"(\d+-\d+-\d+ \d+:\d+:\d+),(\d+) \[\w+\s*\]\s+(.*)", line)
        if m and any(k in m.group(3) for k in ("Start SVIM", "STEP", "Done")):
            t = datetime.datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S").timestamp() + int(m.group(2)) / 1000
            marks.append((t - t0, m.group(3).strip("* ")[:28]))
    ex = [float(l.split()[1]) for l in p.stderr.split("\n") if l.startswith("SVX_EXIT_AT")]
    print("wall %.3f s | " % (t1 - t0) + " | ".join("+%.3f %s" % mk for mk in marks) +
          (" | +%.3f os._exit | +%.3f gone" % (ex[0] - t0, t1 - t0) if ex else ""))
    shutil.rmtree(wd, ignore_errors=True)
