"""Where do the 29 us of the int8 LayerNorm go?  Times the default kernel at the headline shape with parts removed (lab build,
ivit_debug_ln_ablate): 1 no element chain, 2 no statistics, 4 no stores, 8 no table build."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from ivit_amd.prepare import LayerNormParams
DEV = "cuda:0"
rng = np.random.default_rng(0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, C in ((197 * 256, 768), (197 * 256 * 4, 768), (197 * 64, 384)):
    x = torch.from_numpy(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8)).to(DEV)
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32), np.float32(2.0 ** -4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
    out = torch.empty_like(x)
    res = {}
    for bits in (0, 48, 32):
        _lib.call("ivit_debug_ln_ablate", bits)
        res[bits] = round(timeit(lambda: _lib.call("ivit_layernorm_i8", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), C, _lib.stream_ptr())), 1)
    _lib.call("ivit_debug_ln_ablate", 0)
    print(f"rows={rows} C={C}: us by ablation bits {res}", flush=True)
