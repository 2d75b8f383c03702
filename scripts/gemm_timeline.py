"""Per-workgroup timeline of the 256x128 GEMM kernel (diagnostic stamped build): do the two workgroups of a CU
overlap main loop with epilogue?  usage: gemm_timeline.py [fc1|qkv|proj|fc2]"""
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
N, K = shapes[name]
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
b = torch.zeros(N, dtype=torch.int32, device=DEV)
m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV); e = torch.full((N,), 42, dtype=torch.int32, device=DEV)
out = torch.empty(M, N, dtype=torch.int8, device=DEV)
nblk = ((M + 255) // 256) * (N // 128)
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
_lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(stamps))
FL = int(sys.argv[2]) if len(sys.argv) > 2 else 512
_lib.call("ivit_debug_set_gemm_flags", FL)
for _ in range(3):
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, _lib.stream_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nblk, 8)
hw = s[:, 0] & 0xffffffff; xcc = s[:, 0] >> 32
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; wave_slot = hw & 0xf; simd = (hw >> 4) & 3
import time
t0 = s[:, 1].min()
print("span ticks", s[:, 3].max() - t0)
ts, tl, te = s[:, 1] - t0, s[:, 2] - t0, s[:, 3] - t0
tp1, tsy = s[:, 4] - t0, s[:, 5] - t0
print("epilogue split: phase1", (tp1 - tl).mean(), " barrier", (tsy - tp1).mean(), " phase2", (te - tsy).mean())
dr = (s[:, 7] - s[:, 6]).astype(np.float64); dt = (s[:, 3] - s[:, 1]).astype(np.float64)
print("shader clock during the kernel: %.0f MHz (median over workgroups)" % np.median(dt / np.maximum(dr, 1) * 100.0))
print("blocks", nblk, "kernel span (ticks)", te.max(), " mainloop mean", (tl - ts).mean(), " epilogue mean", (te - tl).mean())
key = xcc * 10000 + se * 1000 + sh * 100 + cu
ids = np.unique(key)
print("distinct CUs seen:", len(ids))
# for a few CUs print the timeline sorted by start
for k in ids[:3]:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(ts[idx])]
    print("CU", k, "blocks", len(idx))
    for i in idx[:12]:
        print(f"   blk {i:5d} slot {wave_slot[i]} simd {simd[i]} start {ts[i]:8d} loop_end {tl[i]:8d} end {te[i]:8d}  loop {tl[i]-ts[i]:6d} epi {te[i]-tl[i]:6d}")
gaps = []
for k in ids:
    idx = np.where(key == k)[0]
    for sl in (0, 1):
        j = idx[(wave_slot[idx] & 1) == sl]
        j = j[np.argsort(ts[j])]
        gaps += list(ts[j][1:] - te[j][:-1])
gaps = np.array(gaps)
print("gap between consecutive workgroups on a slot: median %.0f mean %.0f ticks; block total mean %.0f" % (np.median(gaps), gaps.mean(), (te - ts).mean()))
# overlap statistic: fraction of each block's epilogue time during which the co-resident block is in its main loop
tot_epi = 0; tot_cov = 0
for k in ids:
    idx = np.where(key == k)[0]
    for i in idx:
        a0, a1 = tl[i], te[i]
        cov = 0
        for j in idx:
            if j == i: continue
            lo, hi = max(a0, ts[j]), min(a1, tl[j])
            if hi > lo: cov += hi - lo
        tot_epi += a1 - a0; tot_cov += min(cov, a1 - a0)
print("fraction of epilogue time covered by a co-resident main loop:", tot_cov / max(tot_epi, 1))
