"""Stand-alone timing of the DeiT-B row kernels at the headline shape (50 432 rows): LayerNorm int8 (C = 768), ShiftGELU
table form (L = 3072).  Prints microseconds per launch and the HBM rate of the algorithmic bytes (1 B in + 1 B out)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import _lib  # noqa: E402
from ivit_amd.prepare import LayerNormParams  # noqa: E402

DEV = "cuda:0"
rows = 197 * 256
rng = np.random.default_rng(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C in (768, 384, 192):
    x = torch.from_numpy(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8)).to(DEV)
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32),
                         np.float32(2.0 ** -4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
    out = torch.empty_like(x)
    us = timeit(lambda: _lib.call("ivit_layernorm_i8", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e),
                                  _lib.ptr(out), C, _lib.stream_ptr()))
    print(f"layernorm_i8 rows={rows} C={C}: {us:7.1f} us  {2 * rows * C / us / 1e6:6.2f} TB/s", flush=True)

# fused attention, DeiT-B geometry: 256 images x 12 heads x 197 tokens x 64
B, H, T, HD = 256, 12, 197, 64
qkv = torch.from_numpy(np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, HD))), -128, 127).astype(np.int8)).to(DEV)
out = torch.empty(B * T, H * HD, dtype=torch.int8, device=DEV)
from ivit_amd.prepare import dyadic  # noqa: E402
ms, es = dyadic(np.float32(1.0 / 3000.0), np.float32(1.0))
mo, eo = dyadic(np.float32(1.0 / 128.0), np.float32(1.0))
us = timeit(lambda: _lib.call("ivit_attention_fused_i8", _lib.ptr(qkv), _lib.ptr(out), B, H, T, HD, int(ms[0]), int(es[0]), 0.05,
                              int(mo[0]), int(eo[0]), _lib.stream_ptr()))
print(f"attention_fused_i8 B={B} H={H} T={T}: {us:7.1f} us", flush=True)
