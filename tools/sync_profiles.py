#!/usr/bin/env python3
"""Copy the evidence tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<tag>_* , refresh
profiles/traffic.json from the PMC pass and print the numbers the documents quote.

    python3 tools/sync_profiles.py r02
"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for f in ("bench.json", "kernel_stats.csv", "kernel_stats_extras.csv", "bench_under_rocprof.json",
              "kbench_under_rocprof.json", "e2e_scale1.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
    if os.path.exists(os.path.join(src, "pmc", "pmc_summary.json")):
        shutil.copy(os.path.join(src, "pmc", "pmc_summary.json"), os.path.join(dst, "%s_pmc_summary.json" % tag))
        s = json.load(open(os.path.join(dst, "%s_pmc_summary.json" % tag)))
        k = [x for x in s if "cigar_tiles" in x and "4096" in x]
        if k:
            fs, ws = s[k[0]]["FETCH_SIZE"], s[k[0]]["WRITE_SIZE"]
            t = json.load(open(os.path.join(dst, "traffic.json")))
            t["config2_x256_packed"].update({"hbm_bytes_per_launch": int(fs * 1024 * 2 + ws * 1024), "fetch_size_kb": fs,
                                             "write_size_kb": ws, "source": "profiles/%s_pmc_summary.json" % tag})
            json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
            print("traffic", int(fs * 1024 * 2 + ws * 1024))
    if os.path.exists(os.path.join(src, "variants.txt")):
        text = open(os.path.join(src, "variants.txt")).read()
        text = re.sub(r"^-- ", "", text, flags=re.M)
        text = re.sub(r"^--\s+ops/step", "(default)                                 ops/step", text, flags=re.M)
        open(os.path.join(dst, "%s_workload_variants.txt" % tag), "w").write(text)
    r = json.load(open(os.path.join(dst, "%s_bench.json" % tag)))
    rf = r["roofline"]
    print("value %.4e ops/s, %.4f ms/step" % (r["value"], r["ms_per_step"]))
    print("k_cigar_tiles %.1f us (events) -> %.0f GB/s frac %.3f; path %.1f us frac %.3f; traffic %s" % (
        rf["kernel_ms"] * 1e3, rf["achieved"], rf["frac"], rf["path_ms"] * 1e3, rf["path_frac"], rf["traffic"]))
    if "cpu_baseline" in r:
        cb = r["cpu_baseline"]
        print("cpu port %.3g ops/s 1 core, %.3g all cores, cpython %.3g" % (cb["value"], cb.get("all_cores_value", 0), cb.get("cpython_value", 0)))
    for leg in ("latency_case", "roofline_pair", "roofline_editdist", "e2e"):
        v = r.get(leg)
        if not v:
            continue
        if leg == "latency_case":
            print("latency %.1f us/step frac %.3f" % (v["ms_per_step"] * 1e3, v["frac"]))
        elif leg == "roofline_pair":
            print("pair", [(c["candidates"], round(c["ms"] * 1e3, 1)) for c in v["cases"]])
        elif leg == "roofline_editdist":
            print("edit", [(c["pairs"], round(c["ms"], 2), round(c["frac"], 3), round(c["two_stage_plan_ms"], 2)) for c in v["cases"]])
        else:
            print("e2e", round(v["wall_s"], 3), {k: round(x, 3) for k, x in v["phases_s"].items()}, v["oracle_pipeline_wall_s"],
                  v["vcf_identical"], v.get("vcf_matches_real_reference_digest"))
    for f in ("kernel_stats.csv", "kernel_stats_extras.csv"):
        path = os.path.join(dst, "%s_%s" % (tag, f))
        if not os.path.exists(path):
            continue
        print("==", f)
        for x in csv.DictReader(open(path)):
            n = x["Name"]
            if "k_" in n and "at::" not in n:
                print("  %-44s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n.split("k_", 1)[1].split("(")[0][:42], x["Calls"],
                      float(x["AverageNs"]) / 1e3, float(x["MinNs"]) / 1e3, float(x["MaxNs"]) / 1e3))
    e = os.path.join(dst, "%s_e2e_scale1.json" % tag)
    if os.path.exists(e):
        r = json.load(open(e))
        print("full scale", {k: r.get(k) for k in ("open_index_s", "collect_s", "pair_s", "vcf_s", "product_total_s", "all_runs_total_s",
                                                   "cli_wall_s", "oracle_total_s", "vcf_identical")})


if __name__ == "__main__":
    main()
