"""Workgroups per (image, head) of the fused attention kernel against the batch (lab build; bits 28-30 of ivit_debug_ln_ablate
force the split): the launch time for parts = 1, 2, 4 and what attention_parts() picks (0 = its own choice)."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd  # noqa: F401
from ivit_amd import _lib
from ivit_amd.prepare import dyadic
DEV = "cuda:0"
T, HD = 197, 64
ms, es = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -2))
mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))
rng = np.random.default_rng(0)
for B, H in [(1, 3), (1, 12), (16, 12), (32, 12), (64, 6), (64, 12), (96, 12), (128, 12), (160, 12), (256, 12)]:
    qkv = torch.from_numpy(np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, HD))), -128, 127).astype(np.int8)).to(DEV)
    out = torch.empty(B * T, H * HD, dtype=torch.int8, device=DEV)
    res = {}
    for parts in (0, 1, 2, 4):
        _lib.call("ivit_debug_ln_ablate", parts << 28)
        def run():
            _lib.call("ivit_attention_fused_i8", _lib.ptr(qkv), _lib.ptr(out), B, H, T, HD, int(ms[0]), int(es[0]), 0.25, int(mo[0]),
                      int(eo[0]), _lib.stream_ptr())
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[parts] = e0.elapsed_time(e1) / 30 * 1e3
    print(f"B={B:4d} H={H:2d} (B*H={B*H:5d}): auto {res[0]:6.1f}  parts 1 {res[1]:6.1f}  2 {res[2]:6.1f}  4 {res[4]:6.1f} us", flush=True)
_lib.call("ivit_debug_ln_ablate", 0)
