"""Run one DeiT-B GEMM shape (batch 256) a few times: for rocprofv3 --pmc passes.  usage: gemm_one.py fc1 [flags]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
DEV = "cuda:0"
M = 197 * 256
shapes = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
name = sys.argv[1] if len(sys.argv) > 1 else "fc1"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, K = shapes[name]
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
b = torch.zeros(N, dtype=torch.int32, device=DEV)
m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV); e = torch.full((N,), 42, dtype=torch.int32, device=DEV)
out = torch.empty(M, N, dtype=torch.int8, device=DEV)
_lib.call("ivit_debug_set_gemm_flags", flags)
for _ in range(5):
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, _lib.stream_ptr())
torch.cuda.synchronize()
