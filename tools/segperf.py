import os, sys, numpy as np, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from svim_asm_amd import _lib
dev=torch.device('cuda',0)
ctx=_lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.set_timing(True)
for n_reads in (1000, 16000, 64000):
    rng=np.random.default_rng(1)
    k=rng.integers(2,5,size=n_reads); off=np.concatenate(([0],np.cumsum(k))).astype(np.uint32); n=int(off[-1])
    segs=rng.integers(0,1000000,size=(n,6)).astype(np.int32); segs[:,5]&=1
    d_s=torch.from_numpy(segs).to(dev); d_o=torch.from_numpy(off.view(np.int32)).to(dev); d_l=torch.from_numpy(rng.integers(1000,100000,size=n_reads).astype(np.int32)).to(dev)
    d_r=torch.empty((n,8),dtype=torch.int32,device=dev)
    prm=_lib.SegParams(40,100000,50,50,50,50)
    ts=[]
    for i in range(12):
        ctx._check(ctx.lib.svx_segments_classify_dev(ctx.h,d_s.data_ptr(),n,d_o.data_ptr(),n_reads,d_l.data_ptr(),C.byref(prm),d_r.data_ptr()))
        ctx.sync(); t,d=ctx.last_kernel_ms()
        if i>1: ts.append(d)
    print(n_reads, 'kernel_us %.1f'%(np.mean(ts)*1e3))
