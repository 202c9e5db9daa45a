cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do for v in fg4 fg2 fg1; do
SVX_LIB=$PWD/build/libsvx_$v.so python tools/collect_probe.py latency_case product_point 2>/dev/null | python -c "
import json,sys
out=[]
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): out.append('%s %.2f' % (k[:7], v['ms_per_step']*1e3))
print('$v', ' '.join(out))"
done; done
SVX_LIB=$PWD/build/libsvx_fg1.so timeout 600 python -m pytest tests/test_gpu_cigar.py tests/test_gpu_collect.py -x -q 2>&1 | tail -2
