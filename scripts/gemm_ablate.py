"""Time the DeiT-B GEMM shapes (batch 256) with the ablation flags of ivit_debug_set_gemm_flags."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
DEV = "cuda:0"
M = 197 * 256
rng = np.random.default_rng(0)
MODE = os.environ.get("IVIT_DATA", "random")
def rnd(*shape):
    if MODE == "zero": return torch.zeros(shape, dtype=torch.int8, device=DEV)
    if MODE == "small": return torch.from_numpy(rng.integers(-8, 9, size=shape).astype(np.int8)).to(DEV)
    return torch.from_numpy(rng.integers(-128, 128, size=shape).astype(np.int8)).to(DEV)
shapes = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
flags_list = [int(x) for x in (sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "6", "7"])]
for name, (N, K) in shapes.items():
    A, W = rnd(M, K), rnd(N, K)
    b = torch.zeros(N, dtype=torch.int32, device=DEV)
    m = torch.full((N,), 1 << 30, dtype=torch.int32, device=DEV); e = torch.full((N,), 42, dtype=torch.int32, device=DEV)
    out = torch.empty(M, N, dtype=torch.int8, device=DEV)
    st = _lib.stream_ptr()
    for fl in flags_list:
        _lib.call("ivit_debug_set_gemm_flags", fl)
        def run():
            _lib.call("ivit_gemm_i8_requant", _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:5s} N={N:5d} K={K:5d} flags={fl} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TOPS ({2*M*N*K/ms/1e9/5033*100:5.1f}% of peak)", flush=True)
    _lib.call("ivit_debug_set_gemm_flags", 0)
