import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from oracle import oracle as orc
DEV="cuda:0"
KEEP=[]
def dev(a):
    t=torch.from_numpy(np.ascontiguousarray(a)).to(DEV); KEEP.append(t); return t
st=_lib.stream_ptr
rng=np.random.default_rng(0)
for (M,N,K) in [(128,128,64),(128,128,128),(256,128,64)]:
    A=rng.integers(-128,128,size=(M,K)).astype(np.int8); W=rng.integers(-128,128,size=(N,K)).astype(np.int8)
    out=torch.empty(M,N,dtype=torch.int32,device=DEV)
    _lib.call("ivit_gemm_i8_i32",_lib.ptr(dev(A)),K,_lib.ptr(dev(W)),K,None,_lib.ptr(out),N,M,N,K,st())
    got=out.cpu().numpy(); exp=orc.gemm_i8(A,W)
    bad=(got!=exp)
    print("i32",M,N,K,"bad",bad.sum(),"of",bad.size)
    if bad.any():
        print(" bad rows:",np.unique(np.nonzero(bad)[0])[:40]); print(" bad cols:",np.unique(np.nonzero(bad)[1])[:40])
        # try to find where got[t,n] comes from: search exp for matching values
        t,n=np.argwhere(bad)[0]; print(" first bad",t,n,"got",got[t,n],"exp",exp[t,n], "found at", np.argwhere(exp==got[t,n])[:5].tolist())
        # one-hot probes
        A1=np.zeros((M,K),np.int8); A1[5,:]=1
        W1=np.ones((N,K),np.int8)
        _lib.call("ivit_gemm_i8_i32",_lib.ptr(dev(A1)),K,_lib.ptr(dev(W1)),K,None,_lib.ptr(out),N,M,N,K,st())
        g=out.cpu().numpy(); print(" token-row5 probe: nonzero rows",np.unique(np.nonzero(g)[0])[:10],"cols",np.unique(np.nonzero(g)[1])[:10], "val",np.unique(g))
        A1=np.ones((M,K),np.int8); W1=np.zeros((N,K),np.int8); W1[7,:]=1
        _lib.call("ivit_gemm_i8_i32",_lib.ptr(dev(A1)),K,_lib.ptr(dev(W1)),K,None,_lib.ptr(out),N,M,N,K,st())
        g=out.cpu().numpy(); print(" chan-row7 probe: nonzero rows",np.unique(np.nonzero(g)[0])[:10],"cols",np.unique(np.nonzero(g)[1])[:10], "val",np.unique(g))
        A1=np.zeros((M,K),np.int8); A1[:,3]=1; W1=np.zeros((N,K),np.int8); 
        for kk in range(K):
            W1[:]=0; W1[:,kk]=1
            _lib.call("ivit_gemm_i8_i32",_lib.ptr(dev(A1)),K,_lib.ptr(dev(W1)),K,None,_lib.ptr(out),N,M,N,K,st())
            g=out.cpu().numpy()
            if g.any(): print(" A k=3 pairs with W k=",kk,"vals",np.unique(g))
    # requant path
    b=rng.integers(-50000,50000,size=N).astype(np.int32)
    from ivit_amd.prepare import dyadic
    pre=(rng.uniform(0.5,1.0,size=N)*2.0**rng.integers(-16,-9,size=N)).astype(np.float32)
    m,e=dyadic(pre,np.float32(1.0))
    o8=torch.empty(M,N,dtype=torch.int8,device=DEV)
    _lib.call("ivit_gemm_i8_requant",_lib.ptr(dev(A)),K,_lib.ptr(dev(W)),K,_lib.ptr(dev(b)),_lib.ptr(dev(m.view(np.int32))),_lib.ptr(dev(e)),_lib.ptr(o8),N,M,N,K,st())
    exp8=orc.requant(orc.gemm_i8(A,W,b),m.astype(np.float64),e,8)
    bad=(o8.cpu().numpy().astype(np.int32)!=exp8); print("rq",M,N,K,"bad",bad.sum())
