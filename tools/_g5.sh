cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04s; mkdir -p $o; rm -f $o/ab.txt
for r in 1 2 3 4; do for v in edold edpf0 edpf1; do
SVX_LIB=$PWD/build/libsvx_$v.so python tools/kbench.py editdist 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$v', ' '.join('%.3f %.3f' % (c['ms'], c['two_stage_plan_ms']) for c in r['cases']))" >> $o/ab.txt
done; done
sort $o/ab.txt
