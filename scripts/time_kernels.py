"""Time the non-GEMM kernels of the DeiT-B batch-256 forward in isolation (torch events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from ivit_amd.prepare import dyadic, LayerNormParams
DEV = "cuda:0"
B, T, C, H, hd = 256, 197, 768, 12, 64
M = B * T
rng = np.random.default_rng(0)
st = _lib.stream_ptr

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

which = sys.argv[1:] or ["attn", "ln", "gelu"]
if "attn" in which:
    qkv = torch.from_numpy(np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, hd))), -128, 127).astype(np.int8)).to(DEV)
    out = torch.empty(M, C, dtype=torch.int8, device=DEV)
    ms, es = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -2)); mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))
    us = timeit(lambda: _lib.call("ivit_attention_fused_i8", _lib.ptr(qkv), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]), 0.25, int(mo[0]), int(eo[0]), st()))
    print(f"attention  {us:8.1f} us   ({2*2*B*H*T*T*hd/us/1e6:7.1f} TOPS, {4*M*C/us/1e3:6.1f} GB/s algorithmic)")
if "ln" in which:
    x = torch.from_numpy(np.clip(np.rint(rng.normal(0, 30, size=(M, C))), -128, 127).astype(np.int8)).to(DEV)
    out = torch.empty(M, C, dtype=torch.int8, device=DEV)
    lp = LayerNormParams(rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32), np.float32(2.0 ** -5))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    bi, sl, mm, ee = d(lp.bias_int), d(lp.s_ln), d(lp.m.view(np.int32)), d(lp.e)
    us = timeit(lambda: _lib.call("ivit_layernorm_i8", _lib.ptr(x), C, M, C, _lib.ptr(bi), _lib.ptr(sl), _lib.ptr(mm), _lib.ptr(ee), _lib.ptr(out), C, st()))
    print(f"layernorm  {us:8.1f} us   ({2*M*C/us/1e3:6.1f} GB/s algorithmic)")
if "gelu" in which:
    x = torch.from_numpy(np.clip(np.rint(rng.normal(0, 30, size=(M, 4 * C))), -128, 127).astype(np.int8)).to(DEV)
    out = torch.empty(M, 4 * C, dtype=torch.int8, device=DEV)
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    m, e = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -4))
    _lib.call("ivit_shiftgelu_build_lut", 2.0 ** -4, int(m[0]), int(e[0]), _lib.ptr(lut), st())
    us = timeit(lambda: _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(x), 4 * C, M, 4 * C, _lib.ptr(lut), _lib.ptr(out), 4 * C, st()))
    print(f"shiftgelu  {us:8.1f} us   ({2*M*4*C/us/1e3:6.1f} GB/s algorithmic)")
