#!/bin/bash
# round 6: rocprofv3 kernel-trace stats + PMC passes of the device inflate kernels on the 7 261-member SEQ probe (the members a
# full-size run's device leg takes): the shipped form (3: k_inflate_wparse, k_inflate_parse for what the wave parse leaves,
# k_inflate_resolve), the lane-per-member parse (2) and the one-launch kernel (1: k_bgzf_inflate).
# Run on the GPU box: bash tools/r06_inflate_evidence.sh [3|2|1]  -> gpurun_out/r06_infl[_2|_1]/*
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
k=${1:-3}; export SVX_INFLATE_KERNEL=$k
out=gpurun_out/r06_infl; [ "$k" != 3 ] && out=gpurun_out/r06_infl_$k; mkdir -p $out
d=/tmp/svx_infl_ds; mkdir -p $d
args="tools/gpu_inflate_probe.py --scale 0.25 --dataset $d --members 7261 --min-payload 8192 --counts 1000,3000,7261"
python3 $args > $out/probe.json 2> $out/probe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $args > $out/kt_probe.json 2>> $out/probe.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -o p -- python3 $args > /dev/null 2>> $out/probe.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc2 -o p -- python3 $args > /dev/null 2>> $out/probe.err
python3 - $out <<'PY'
import csv, collections, sys, json, glob
out=sys.argv[1]
res={}
for f in glob.glob(out+"/kt/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Name"] or "gather" in r["Name"]:
            res.setdefault("kernel_stats", []).append({k: r[k] for k in ("Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs")})
for p in ("pmc1","pmc2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+"/"+p+"/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for name in ("k_bgzf_inflate", "k_inflate_wparse", "k_inflate_parse", "k_inflate_resolve"):
                if name in r["Kernel_Name"]:
                    agg[name][r["Counter_Name"]].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]), float(r["Counter_Value"])))
    # the launch with the most members = the longest one
    for name, cs in agg.items():
        for k,v in cs.items():
            v.sort(); res.setdefault("pmc_longest_launch", {}).setdefault(name, {})[k]=v[-1][1]; res.setdefault("pmc_longest_launch_ns", {}).setdefault(name, {})[k]=v[-1][0]
try: res["probe"]=json.load(open(out+"/probe.json"))
except Exception as e: res["probe_error"]=str(e)
json.dump(res, open(out+"/summary.json","w"), indent=1)
print(json.dumps(res)[:4000])
PY
