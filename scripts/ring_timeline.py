import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
DEV="cuda:0"; M=197*256
name=sys.argv[1]; N,K={"qkv":(2304,768),"fc2":(768,3072)}[name]
rng=np.random.default_rng(0)
A=torch.from_numpy(rng.integers(-128,128,size=(M,K)).astype(np.int8)).to(DEV); W=torch.from_numpy(rng.integers(-128,128,size=(N,K)).astype(np.int8)).to(DEV)
b=torch.zeros(N,dtype=torch.int32,device=DEV); m=torch.full((N,),(1<<30)+12345,dtype=torch.int32,device=DEV); e=torch.full((N,),42,dtype=torch.int32,device=DEV)
out=torch.empty(M,N,dtype=torch.int8,device=DEV)
stamps=torch.zeros(65536+256*4*8,dtype=torch.int64,device=DEV)
_lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(stamps))
_lib.call("ivit_debug_set_gemm_flags", 8192|32768)
for _ in range(3):
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(A),K,_lib.ptr(W),K,_lib.ptr(b),_lib.ptr(m),_lib.ptr(e),_lib.ptr(out),N,M,N,K,_lib.stream_ptr())
torch.cuda.synchronize()
allst=stamps.cpu().numpy(); s=allst[:65536].reshape(256,4,16,4); tl=allst[65536:].reshape(256,4,8)
for blk in (0,1,100,255):
    t=tl[blk,0]
    print(f'block {blk}: tile start->loop {t[1]-t[0]}, loop {t[2]-t[1]}, sync+prefetch issue {t[3]-t[2]}, epilogue {t[4]-t[3]}, final sync {t[5]-t[4]}, total {t[5]-t[0]}')
nk=min(K//64,16)
for blk in (0,1,100):
  for w in (0,3):
    t=s[blk,w,:nk]
    print(f"block {blk} wave {w}: per-step [batch0+dma, wait+barrier, batch1] and step total")
    for kt in range(nk):
        a0,a1,a2,a3=t[kt]
        nxt=t[kt+1][0] if kt+1<nk else a3
        print(f"  kt={kt:2d}  {a1-a0:6d} {a2-a1:6d} {a3-a2:6d}   total {a3-a0:6d}  gap->next {nxt-a3:5d}")
