#!/bin/bash
# Collect PMC counters of the bench kernels in separate passes (rocprofv3 --pmc; no sys/hip tracing).
# usage: tools/pmc.sh <outdir> [bench args...]          PMC_CMD="tools/collect_probe.py product_point" tools/pmc.sh <outdir>
# (PMC_CMD: another python program instead of bench.py; the program itself follows rocprofv3's `--`, no wrapper)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "$PMC_CMD" ]; then
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out" -o pass$i -- python3 $PMC_CMD > /dev/null 2>> "$out/err.txt"
  else
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out" -o pass$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2>> "$out/err.txt"
  fi
done
python3 - "$out" <<'PY'
import csv, collections, glob, json, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", k)
        if m:
            short = m.group(1) + (m.group(2) or "")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[short]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in res.items():
    print(k, json.dumps({c: round(v) for c, v in sorted(d.items())}))
PY
