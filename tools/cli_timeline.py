#!/usr/bin/env python3
"""Where the wall-clock and the CPU seconds of one fresh `svim-asm diploid` go (GPU box):

    python tools/cli_timeline.py DIR_WITH_hap1.bam_hap2.bam_ref.fa [repeats] [NAME=VALUE ...]

Runs the command `repeats` times (default 5) as fresh processes with SVX_CLI_TIMELINE set (svim_asm_amd/_timeline.py: the
command notes wall time and process CPU seconds at its phase boundaries and writes them when it leaves) and prints, per
mark, the median offset from the parent's Popen and the CPU seconds spent since the previous mark; the stage clocks of
COLLECT / PAIR / VCF of the median run; and the wall-clock of every run.  NAME=VALUE pairs go into the command's environment
(A/B of SVX_* switches)."""
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def throttled():
    """(nr_throttled, throttled_usec) of this cgroup (cpu.stat), or None"""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def main():
    d = sys.argv[1]
    rest = sys.argv[2:]
    env_extra = dict(a.split("=", 1) for a in rest if "=" in a)
    nums = [a for a in rest if "=" not in a]
    reps = int(nums[0]) if nums else 5
    runs = []
    for _ in range(reps):
        wd = tempfile.mkdtemp(prefix="svx_tl_")
        tl = os.path.join(wd, "timeline.jsonl")
        env = dict(os.environ, SVX_CLI_TIMELINE=tl, **env_extra)
        thr0 = throttled()
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "svim-asm"), "diploid", wd, os.path.join(d, "hap1.bam"),
                            os.path.join(d, "hap2.bam"), os.path.join(d, "ref.fa")], env=env, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True)
        t1 = time.time()
        marks = [json.loads(l) for l in open(tl)] if os.path.exists(tl) else []
        thr1 = throttled()
        runs.append({"wall": t1 - t0, "rc": p.returncode, "marks": [dict(m, t=m["t"] - t0) for m in marks],
                     "throttled_ms": None if thr0 is None or thr1 is None else (thr1[1] - thr0[1]) / 1e3,
                     "throttled_periods": None if thr0 is None or thr1 is None else thr1[0] - thr0[0]})
        shutil.rmtree(wd, ignore_errors=True)
    runs.sort(key=lambda r: r["wall"])
    med = runs[len(runs) // 2]
    names = [m["name"] for m in med["marks"]]
    print("wall-clock of %d fresh processes: %s  (median %.3f s)%s" % (reps, " ".join("%.3f" % r["wall"] for r in runs), med["wall"],
                                                                       "  env " + repr(env_extra) if env_extra else ""))
    print("%-26s %10s %10s %12s  %s" % ("mark", "+wall s", "delta s", "cpu since", "thread"))
    prev_t, prev_cpu = 0.0, 0.0
    for k, name in enumerate(names):
        ts = [r["marks"][k]["t"] for r in runs if len(r["marks"]) == len(names) and r["marks"][k]["name"] == name]
        cs = [r["marks"][k]["cpu"] for r in runs if len(r["marks"]) == len(names) and r["marks"][k]["name"] == name]
        t, c = statistics.median(ts), statistics.median(cs)
        main_thread = med["marks"][k]["thread"] == "MainThread"
        print("%-26s %10.3f %10.3f %12.3f  %s" % (name, t, (t - prev_t) if main_thread else float("nan"), c - prev_cpu if main_thread else float("nan"),
                                               med["marks"][k]["thread"]))
        if main_thread:
            prev_t, prev_cpu = t, c
    print("%-26s %10.3f %10.3f" % ("process gone (parent)", med["wall"], med["wall"] - prev_t))
    print("per run (wall | " + " | ".join(n for n in names if n in ("device context created", "files open", "COLLECT done", "PAIR done", "VCF written", "leaving")) + "):")
    for r in runs:
        byname = {m["name"]: m["t"] for m in r["marks"]}
        print("  %.3f | %s" % (r["wall"], " | ".join("%.3f" % byname.get(n, float("nan")) for n in names
                                                  if n in ("device context created", "files open", "COLLECT done", "PAIR done", "VCF written", "leaving"))))
    def gap(r, a_, b_):
        byname = {m["name"]: m["t"] for m in r["marks"]}
        return byname.get(b_, float("nan")) - byname.get(a_, float("nan"))
    gaps = sorted(gap(r, "device context created", "COLLECT done") for r in runs)
    tails = sorted(gap(r, "COLLECT done", "leaving") for r in runs)
    print("device context -> COLLECT done: median %.3f s (%s); COLLECT done -> leaving: median %.3f s" % (
        gaps[len(gaps) // 2], " ".join("%.3f" % g for g in gaps), tails[len(tails) // 2]))
    print("cgroup throttling per run (wall s: periods throttled, ms): %s" % "  ".join(
        "%.3f: %s, %s" % (r["wall"], r["throttled_periods"], "-" if r["throttled_ms"] is None else "%.0f" % r["throttled_ms"]) for r in runs))
    tails = sorted(r["wall"] - r["marks"][-1]["t"] for r in runs if r["marks"])
    print("last mark -> process gone, per run: %s  (median %.3f s)" % (" ".join("%.3f" % x for x in tails), tails[len(tails) // 2]))
    for m in med["marks"]:
        if "stages" in m:
            print("  %s: %s" % (m["name"], " ".join("%s=%.3f" % (k, v) for k, v in sorted(m["stages"].items()) if isinstance(v, (int, float)))))
    total_cpu = statistics.median([r["marks"][-1]["cpu"] for r in runs if r["marks"]])
    print("CPU seconds of the process (all threads) when it leaves: %.2f" % total_cpu)


if __name__ == "__main__":
    main()
