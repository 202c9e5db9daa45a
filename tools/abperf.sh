for lib in "$@"; do echo "== $lib"; SVX_LIB=$PWD/$lib timeout 300 python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from svim_asm_amd import _lib, synth
b = synth.concat_batches([synth.synth_cigar_batch(seed=1200+i) for i in range(8)]*8)
dev = torch.device('cuda',0)
n_ops=int(b['aln_off'][-1]); n_aln=len(b['aln_off'])-1
ctx=_lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
d_c=torch.from_numpy(b['cigar'].view(np.int32)).to(dev); d_off=torch.from_numpy(b['aln_off'].astype(np.int64)).to(dev); d_rs=torch.from_numpy(b['ref_start']).to(dev)
cap=n_ops//16
outs=[torch.empty(cap,dtype=torch.int32,device=dev) for _ in range(4)]+[torch.empty(cap,dtype=torch.uint8,device=dev)]
d_n=torch.zeros(1,dtype=torch.int64,device=dev)
ptrs=tuple(o.data_ptr() for o in outs)
ctx.set_timing(True)
ks=[]
for i in range(25):
    ctx.cigar_extract_dev(d_c.data_ptr(), n_ops, d_off.data_ptr(), n_aln, d_rs.data_ptr(), 40, ptrs, cap, d_n.data_ptr())
    ctx.sync(); t,d=ctx.last_kernel_ms()
    if i>=5: ks.append((t,d))
print('kernel_ms %.4f path_ms %.4f  GB/s %.0f' % (np.mean([k[1] for k in ks]), np.mean([k[0] for k in ks]), 4*n_ops/np.mean([k[1] for k in ks])/1e6))
PY
done
