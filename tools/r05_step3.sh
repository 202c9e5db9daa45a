#!/bin/bash
# round 5, step 3: edit-distance cut-off between strips (parity, then the edit-distance leg), inflate chain floor
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_s3; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_editdist.py -x -q -m gpu > $out/pytest_edit.txt 2>&1; tail -3 $out/pytest_edit.txt
timeout 600 python3 tools/kbench.py editdist > $out/editdist.json 2> $out/editdist.err
python3 -c "
import json; r=json.loads(open('$out/editdist.json').read().strip().splitlines()[-1])
for c in r['cases']: print(c['workload'][:60], 'ms %.2f frac %.3f plan_ms %.2f' % (c['ms'], c['frac'], c['two_stage_plan_ms']))"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/inflate_chain tools/ubench/inflate_chain.hip && /tmp/inflate_chain > $out/inflate_chain.txt 2>&1; cat $out/inflate_chain.txt
