"""A/B of the int8 LayerNorm kernel forms at the shapes of the BASELINE configs (lab build): the round-2/3 grouped kernel
(form 3) against the streaming kernel of ln_stream.h and its lab variants (ring depth, workgroups per CU), plain and COMPAT
(natural input scale), row-major and block-layout output.  Device-scope HIP events on the launch stream, interleaved rounds."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib, hiptime
from ivit_amd.prepare import LayerNormParams, phi_tables
DEV = "cuda:0"
rng = np.random.default_rng(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)

def time_us(fn, n=40):
    st = _lib.stream_ptr()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = hiptime.Event(), hiptime.Event()
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3

shapes = [(197 * 256, 768), (197 * 64, 384), (197 * 128, 768), (197 * 1024, 768), (197, 192)]
if len(sys.argv) > 1 and sys.argv[1] == "--headline":
    shapes = shapes[:1]
variants = [("grouped(r3)", 3, 0, 0), ("stream", 0, 0, 0), ("stream equal-prio", 0, 0, 32), ("stream ng2", 0, 4, 0), ("stream dbuf", 0, 5, 0),
            ("stream 5 wg/CU", 0, 7 | (5 << 4), 0), ("stream 6 wg/CU", 0, 6 | (6 << 4), 0)]
if len(sys.argv) > 1 and sys.argv[1] == "--small":     # launches below ~12 MB: the half-wave kernel with 4 / 2 / 1 row pairs per wave (lab bits 21-22)
    shapes = [(197 * 64, 384), (197 * 16, 384), (197, 192), (197 * 8, 192)]
    variants = [("grouped(r3)", 3, 0, 0), ("product", 0, 0, 0), ("half-wave 4 pairs", 0, 0, 1 << 21), ("half-wave 2 pairs", 0, 0, 2 << 21),
                ("half-wave 1 pair", 0, 0, 3 << 21), ("stream", 4, 0, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "--first":
    shapes = shapes[:1]
for rows, C in shapes:
    x = t(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8))
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32), np.float32(2.0 ** -4))
    b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
    remap, phi = phi_tables(np.float32(0.0371))
    remap, phi = t(remap), t(phi)
    out = torch.empty((rows + 15) // 16 * 16, C, dtype=torch.int8, device=DEV)
    ref = {}
    for compat in (0, 1):
        for blocks in (1, 0):
            def call():
                if compat:
                    _lib.call("ivit_layernorm_i8_compat", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e),
                              _lib.ptr(remap), _lib.ptr(phi), _lib.ptr(out), C, blocks, _lib.stream_ptr())
                else:
                    _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e),
                              _lib.ptr(out), C, blocks, _lib.stream_ptr())
            res = {}
            for rnd in range(2):
                for name, form, cfg, ablbits in variants:
                    if cfg and C != 768 and (cfg & 15):
                        continue
                    _lib.call("ivit_debug_ln_wave_per_row", form)
                    _lib.call("ivit_debug_ln_stream_cfg", cfg)
                    _lib.call("ivit_debug_ln_ablate", ablbits)
                    out.zero_()
                    us = time_us(call)
                    res[name] = min(res.get(name, 1e9), us)
                    got = out.clone()
                    key = (compat, blocks)
                    if key not in ref:
                        ref[key] = got
                    elif not torch.equal(ref[key], got):
                        print(f"   MISMATCH {name} vs grouped: {(ref[key] != got).sum().item()} bytes", flush=True)
            _lib.call("ivit_debug_ln_wave_per_row", 0); _lib.call("ivit_debug_ln_stream_cfg", 0); _lib.call("ivit_debug_ln_ablate", 0)
            mb = 2 * rows * C / 1e6
            print(f"rows={rows} C={C} compat={compat} blocks={blocks} ({mb:.1f} MB): " +
                  "  ".join(f"{k} {v:.1f}us ({mb / v / 1e3 * 1e3:.2f} TB/s)" if False else f"{k} {v:.1f}" for k, v in res.items()), flush=True)
