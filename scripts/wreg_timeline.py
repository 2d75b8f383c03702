"""Stamped timeline of the weights-in-registers GEMM (lab flag 16; +4096: one workgroup per CU): shader cycles per tile phase
and per K step, wave 0 of a few workgroups.  usage: wreg_timeline.py [qkv|proj|fc1|fc2] [one]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd  # noqa: F401
from ivit_amd import _lib
DEV = "cuda:0"; M = 197 * 256
name = sys.argv[1] if len(sys.argv) > 1 else "fc1"
N, K = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}[name]
one = "one" in sys.argv
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
At, Wf = torch.empty_like(A), torch.empty_like(W)
_lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), _lib.stream_ptr())
_lib.call("ivit_pack_weight_frags_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), _lib.stream_ptr())
b = torch.zeros(N, dtype=torch.int32, device=DEV); m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV)
e = torch.full((N,), 42, dtype=torch.int32, device=DEV); out = torch.empty(M, N, dtype=torch.int8, device=DEV)
stamps = torch.zeros(512 * 4 * 32, dtype=torch.int64, device=DEV)
_lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(stamps))
_lib.call("ivit_debug_set_gemm_flags", 16 | (4096 if one else 0))
for _ in range(3):
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, 9, _lib.stream_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(512, 4, 32)
nb = 256 if one else 512
for it in (0, 1):
    t = s[:nb, it]
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
    print(f"tile {it}: median over workgroups  start->loop {np.median(d[:,0]):.0f}  loop {np.median(d[:,1]):.0f}  epilogue {np.median(d[:,2]):.0f}  total {np.median(d[:,3]):.0f}")
    if it + 1 < 4:
        print(f"   tile start to next tile start {np.median(s[:nb, it + 1, 0] - t[:, 0]):.0f}")
dt = (s[:nb, 3, 0] - s[:nb, 0, 0]).astype(np.float64)
dr = (s[:nb, 3, 16] - s[:nb, 0, 16]).astype(np.float64)
print(f"s_memtime ticks per s_memrealtime tick (100 MHz): median {np.median(dt / dr):.2f}  -> counter rate {np.median(dt / dr) * 100:.0f} MHz")
for blk in (0, 1, 100):
    t = s[blk, 1]
    steps = np.diff(t[4:16])
    print(f"workgroup {blk} tile 1: K-step durations {steps.tolist()}  loop {t[2] - t[1]}  epilogue {t[3] - t[2]}")
