#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=$(python3 -c "
import tempfile,sys
sys.path.insert(0,'.')
from svim_asm_amd import synth_bam
from tools import e2e_bench
d=tempfile.mkdtemp(prefix='svx_ds_'); synth_bam.write_dataset(d, **e2e_bench.dataset_args(1.0)); print(d)" 2>/dev/null | tail -1)
for i in 1 2; do python3 tools/cohort_timeline.py $d 8; done
python3 tools/cohort_timeline.py $d 8 --cohort_workers 6
python3 tools/cli_timeline.py $d 5
python3 tools/cli_timeline.py $d 5 SVX_BAM_DEVICE_INFLATE=50
python3 tools/cli_timeline.py $d 5 SVX_BAM_DEVICE_INFLATE=100
python3 tools/cli_timeline.py $d 5 OPENBLAS_NUM_THREADS=64
