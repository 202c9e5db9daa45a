#!/bin/bash
# PMC counters of the pair sort kernels (tools/pairbench.py 60000: k_pair_single in both key orders, and the radix
# plan's kernels), separate rocprofv3 --pmc passes, kernel-trace only.  usage: tools/pmc_pair.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out" -o pass$i -- python3 tools/pairbench.py 60000 > /dev/null 2>> "$out/err.txt"
done
python3 - "$out" <<'PY'
import csv, collections, glob, json, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            short = m.group(1) + (m.group(2) or "")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[short]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/pmc_pair_summary.json", "w"), indent=1, sort_keys=True)
for k, d in res.items():
    print(k, json.dumps({c: round(v) for c, v in sorted(d.items())}))
PY
rm -f "$out"/pass*_kernel_trace.csv "$out"/pass*_agent_info.csv
