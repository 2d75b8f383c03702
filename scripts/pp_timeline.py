"""Time stamps (s_memtime) of the ping-pong GEMM kernel's tile phases: K loop / barrier / phase 2 / copy / barrier.
usage: pp_timeline.py qkv|fc2 [extra_flag_bits]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import _lib  # noqa: E402

DEV = "cuda:0"
M = 197 * 256
name = sys.argv[1]
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, K = {"qkv": (2304, 768), "fc2": (768, 3072), "proj": (768, 768), "fc1": (3072, 768)}[name]
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
b = torch.zeros(N, dtype=torch.int32, device=DEV)
m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV)
e = torch.full((N,), 42, dtype=torch.int32, device=DEV)
out = torch.empty(M, N, dtype=torch.int8, device=DEV)
stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device=DEV)
_lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(stamps))
_lib.call("ivit_debug_set_gemm_flags", 8388608 | 33554432 | extra)
for _ in range(3):
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N,
              K, _lib.stream_ptr())
torch.cuda.synchronize()
_lib.call("ivit_debug_set_gemm_flags", 0)
t = stamps.cpu().numpy().reshape(256, 2, 8)
for blk in (0, 1, 100, 255):
    for it in (0, 1):
        s = t[blk, it]
        print(f"block {blk:3d} tile {it + 1}: K loop {s[1] - s[0]:6d}  sync {s[2] - s[1]:5d}  table+phase2 {s[3] - s[2]:5d}  "
              f"copy {s[4] - s[3]:5d}  sync {s[5] - s[4]:5d}   total {s[5] - s[0]:6d}")
d = t[:, :, 1:6] - t[:, :, 0:5]
print("mean over blocks:", d.reshape(-1, 5).mean(axis=0).round(0), "total", (t[:, :, 5] - t[:, :, 0]).mean().round(0))
