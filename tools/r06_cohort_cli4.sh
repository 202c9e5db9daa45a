#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=$(python3 -c "
import tempfile,sys
sys.path.insert(0,'.')
from svim_asm_amd import synth_bam
from tools import e2e_bench
d=tempfile.mkdtemp(prefix='svx_ds_'); synth_bam.write_dataset(d, **e2e_bench.dataset_args(1.0)); print(d)" 2>/dev/null | tail -1)
python3 tools/cohort_timeline.py $d 8
for n in 8 16; do for kt in "" "--cohort_workers 6 --cohort_threads 2" "--cohort_workers 3 --cohort_threads 4"; do
  echo "== N=$n $kt"; python3 tools/cohort_timeline.py $d $n $kt | head -1
  echo "== N=$n $kt (no buffer cache)"; SVX_BAM_NO_BUFFER_CACHE=1 python3 tools/cohort_timeline.py $d $n $kt | head -1
done; done
echo "== N=32"; python3 tools/cohort_timeline.py $d 32 | head -1
python3 tools/cli_timeline.py $d 7
