#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=$(python3 -c "
import tempfile,sys
sys.path.insert(0,'.')
from svim_asm_amd import synth_bam
from tools import e2e_bench
d=tempfile.mkdtemp(prefix='svx_ds_'); synth_bam.write_dataset(d, **e2e_bench.dataset_args(1.0)); print(d)" 2>/dev/null | tail -1)
for v in "" "SVX_INGEST_THREADS=8" "SVX_INGEST_THREADS=16" "SVX_BAM_DEVICE_INFLATE=60" "SVX_BAM_DEVICE_INFLATE=60 SVX_INGEST_THREADS=8" "SVX_BAM_DEVICE_INFLATE=100 SVX_INGEST_THREADS=8"; do
  echo "== $v"
  python3 tools/cli_timeline.py $d 9 $v | grep -E "wall-clock|COLLECT done  |PAIR done  |last mark|throttling|CPU seconds"
done
