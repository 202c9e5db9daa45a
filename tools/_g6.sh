cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in inf_l8 inf_l16; do SVX_LIB=$PWD/build/libsvx_$v.so timeout 600 python tools/gpu_inflate_probe.py --scale 0.25 --members 20000 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['device_kernel_ms'], r['all_status_ok_and_bytes_equal_zlib_on_sample'])"; done
