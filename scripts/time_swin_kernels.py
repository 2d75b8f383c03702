"""Time the Swin-specific kernels in isolation at the Swin-T batch-128 shapes (events on the launch stream)."""
import os
import sys

if {"lnc", "pn", "gelu"} & set(sys.argv[1:]):
    os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import _lib  # noqa: E402
from ivit_amd.prepare import LayerNormParams, dyadic  # noqa: E402

DEV = "cuda:0"
rng = np.random.default_rng(0)
st = _lib.stream_ptr
B = int(os.environ.get("IVIT_B", "128"))


def d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


which = sys.argv[1:] or ["ln", "attn", "res"]
STAGES = [(56, 96, 3), (28, 192, 6), (14, 384, 12), (7, 768, 24)]
if "ln" in which:
    for H, C, nH in STAGES + [(28, 384, 0), (14, 768, 0), (7, 1536, 0)]:
        rows = B * H * H
        x = d(rng.integers(-20000, 20000, size=(rows, C)).astype(np.int16))
        out = torch.empty(rows, C, dtype=torch.int8, device=DEV)
        lp = LayerNormParams(rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32), np.float32(2.0 ** -5))
        bi, sl, mm, ee = d(lp.bias_int), d(lp.s_ln), d(lp.m.view(np.int32)), d(lp.e)
        for ws, sh in ((0, 0), (7, 3)) if nH else ((0, 0),):
            us = timeit(lambda: _lib.call("ivit_layernorm_i16_i8", _lib.ptr(x), rows, C, _lib.ptr(bi), _lib.ptr(sl), _lib.ptr(mm),
                                          _lib.ptr(ee), _lib.ptr(out), C, H, H, ws, sh if H > 7 else 0, st()))
            print(f"ln16 rows={rows:7d} C={C:5d} ws={ws} shift={sh}: {us:8.1f} us  ({3 * rows * C / us / 1e3:7.1f} GB/s algorithmic)", flush=True)
if "lnc" in which:     # natural-scale 16-bit LayerNorm: registers (product) against the round-3 LDS sums (lab bit 20), interleaved
    os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
    for H, C, nH in STAGES:
        rows = B * H * H
        x = d(rng.integers(-20000, 20000, size=(rows, C)).astype(np.int16))
        out = torch.empty(rows, C, dtype=torch.int8, device=DEV)
        lp = LayerNormParams(rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32), np.float32(2.0 ** -5))
        bi, sl, mm, ee = d(lp.bias_int), d(lp.s_ln), d(lp.m.view(np.int32)), d(lp.e)
        res = {}
        for rnd in range(5):
            for form in (0, 1 << 20):
                _lib.call("ivit_debug_ln_ablate", form)
                us = timeit(lambda: _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(x), rows, C, 0.000913, 1, _lib.ptr(bi), _lib.ptr(sl),
                                              _lib.ptr(mm), _lib.ptr(ee), _lib.ptr(out), C, 0, 0, 0, 0, st()), n=10)
                res.setdefault(form, []).append(us)
        _lib.call("ivit_debug_ln_ablate", 0)
        us = timeit(lambda: _lib.call("ivit_layernorm_i16_i8", _lib.ptr(x), rows, C, _lib.ptr(bi), _lib.ptr(sl), _lib.ptr(mm),
                                      _lib.ptr(ee), _lib.ptr(out), C, 0, 0, 0, 0, st()), n=10)
        if C == 96:      # stage 0: the transposed-view order (every LayerNorm of the stage)
            ro = {}
            for rnd in range(5):
                for form in (0, 1 << 20):
                    _lib.call("ivit_debug_ln_ablate", form)
                    ro.setdefault(form, []).append(timeit(lambda: _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(x), rows, C, 0.000913, 1 | ((H * H) << 8),
                                                                            _lib.ptr(bi), _lib.ptr(sl), _lib.ptr(mm), _lib.ptr(ee), _lib.ptr(out), C, 0, 0, 0, 0, st()), n=10))
            _lib.call("ivit_debug_ln_ablate", 0)
            print(f"ln16 natural rows={rows:7d} C={C:4d} outer-reduction order: registers {np.median(ro[0]):7.1f} us   LDS sums {np.median(ro[1 << 20]):7.1f} us", flush=True)
        print(f"ln16 natural rows={rows:7d} C={C:4d}: registers {np.median(res[0]):7.1f} us   LDS sums {np.median(res[1 << 20]):7.1f} us   "
              f"(power-of-two kernel {us:7.1f} us; {3 * rows * C / np.median(res[0]) / 1e3:7.1f} GB/s algorithmic)", flush=True)
if "pn" in which:      # the patch norm (8-bit LayerNorm over 96 channels): one-dword half-wave kernel against the two-dword one (lab bits 25 / 26)
    rows, C = B * 56 * 56, 96
    x = d(rng.integers(-128, 128, size=(rows, C)).astype(np.int8))
    out = torch.empty(rows, C, dtype=torch.int8, device=DEV)
    lp = LayerNormParams(rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32), np.float32(2.0 ** -5))
    bi, sl, mm, ee = d(lp.bias_int), d(lp.s_ln), d(lp.m.view(np.int32)), d(lp.e)
    res, outs = {}, {}
    for rnd in range(5):
        for form in (0, 1 << 27, 1 << 26, 1 << 25):
            _lib.call("ivit_debug_ln_ablate", form)
            res.setdefault(form, []).append(timeit(lambda: _lib.call("ivit_layernorm_i8", _lib.ptr(x), C, rows, C, _lib.ptr(bi), _lib.ptr(sl), _lib.ptr(mm),
                                                                     _lib.ptr(ee), _lib.ptr(out), C, st()), n=10))
            outs[form] = out.clone()
    _lib.call("ivit_debug_ln_ablate", 0)
    assert all(torch.equal(outs[0], o) for o in outs.values())
    print(f"patch norm rows={rows} C={C}: one dword, 8 row pairs {np.median(res[0]):7.1f} us   one dword, 4 row pairs {np.median(res[1 << 26]):7.1f} us   ds_bpermute row sums {np.median(res[1 << 27]):7.1f} us   "
          f"two dwords (round 3) {np.median(res[1 << 25]):7.1f} us   ({2 * rows * C / np.median(res[0]) / 1e3:7.1f} GB/s algorithmic)", flush=True)
if "gelu" in which:    # ShiftGELU table pass, row-major rows (the Swin MLP) and the ViT shapes in both layouts; lab bits 24 / 28: whole wave per short row / prefetch of the next rows
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    mg, eg = dyadic(np.float32(0.05 * 0.05 / 128), np.float32(0.02))
    _lib.call("ivit_shiftgelu_build_lut", 0.05, int(mg[0]), int(eg[0]), _lib.ptr(lut), st())
    for rows, L in [(B * 3136, 384), (B * 784, 768), (B * 196, 1536), (B * 49, 3072), (64 * 197, 1536), (256 * 197, 3072)]:
        x = d(np.clip(np.rint(rng.normal(0, 30, size=(rows, L))), -128, 127).astype(np.int8))
        out = torch.empty_like(x)
        line = f"ShiftGELU table pass rows={rows:7d} L={L:5d}:"
        forms = [("row-major", 0, 0)] + ([("with prefetch", 0, 1 << 28), ("wave per row", 0, 1 << 24)] if L <= 384 else []) + ([("block layout", 3, 0)] if L % 64 == 0 else [])
        res = {}
        for rnd in range(5):
            for name, lay, bits in forms:
                _lib.call("ivit_debug_ln_ablate", bits)
                res.setdefault(name, []).append(timeit(lambda: _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(x), L, rows, L, _lib.ptr(lut), _lib.ptr(out), L, lay, st()), n=10))
        _lib.call("ivit_debug_ln_ablate", 0)
        for name, _, _ in forms:
            us = np.median(res[name])
            line += f"   {name} {us:6.1f} us ({2 * rows * L / us / 1e6:4.2f} TB/s)"
        print(line, flush=True)
if "attn" in which:
    for H, C, nH in STAGES:
        nwin = B * (H // 7) ** 2
        qkv = d(np.clip(np.rint(rng.normal(0, 40, size=(3, nwin, nH, 49, 32))), -128, 127).astype(np.int8))
        out = torch.empty(nwin * 49, C, dtype=torch.int8, device=DEV)
        bias = d(rng.integers(-60, 61, size=(nH, 49, 64)).astype(np.int16))
        mask = d(rng.integers(0, 3, size=((H // 7) ** 2, 64)).astype(np.uint8))
        ms, es = dyadic(np.float32(2.0 ** -9), np.float32(0.25))
        mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))
        for mk in (None, mask):
            us = timeit(lambda: _lib.call("ivit_window_attention_i8", _lib.ptr(qkv), _lib.ptr(out), C, _lib.ptr(bias), _lib.ptr(mk), -400, nwin,
                                          (H // 7) ** 2, nH, 49, 32, int(ms[0]), int(es[0]), 1 << 30, 30, 0.25, int(mo[0]), int(eo[0]), st()))
            print(f"window attention windows={nwin:5d} heads={nH:2d} mask={mk is not None}: {us:8.1f} us  "
                  f"({4 * nwin * 49 * C / us / 1e3:7.1f} GB/s algorithmic, {4 * nwin * nH * 49 * 49 * 32 / us / 1e6:6.1f} TOPS)", flush=True)
if "attnc" in which:     # natural-scale window attention: literal float Shiftmax (round 3) against the band table (round 4); power-of-two beside
    from ivit_amd.prepare import window_shiftexp_band
    s_at = np.float32(0.271)
    qv = np.arange(-128, 128, dtype=np.float32)
    phi = d(((qv * s_at).astype(np.float32) / s_at).astype(np.float32))
    phim = d(((((qv * s_at).astype(np.float32) + np.float32(-100.0)).astype(np.float32)) / s_at).astype(np.float32))
    band, bw = window_shiftexp_band(s_at, True)
    bandd = d(band)                                     # one row at this scale
    band256 = d(np.repeat(band, 256, axis=0)) if band.shape[0] == 1 else bandd
    for H, C, nH in STAGES:
        nwin = B * (H // 7) ** 2
        qkv = d(np.clip(np.rint(rng.normal(0, 40, size=(3, nwin, nH, 49, 32))), -128, 127).astype(np.int8))
        out = torch.empty(nwin * 49, C, dtype=torch.int8, device=DEV)
        bias = d(rng.integers(-60, 61, size=(nH, 49, 64)).astype(np.int16))
        mask = d(rng.integers(0, 3, size=((H // 7) ** 2, 64)).astype(np.uint8))
        ms, es = dyadic(np.float32(2.0 ** -9), s_at)
        mb, eb = dyadic(s_at * np.float32(0.75), s_at)
        mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))
        r = {"lit": [], "band": [], "band256": [], "pow2": []}
        for rnd in range(5):
            r["lit"].append(timeit(lambda: _lib.call("ivit_window_attention_i8_compat", _lib.ptr(qkv), _lib.ptr(out), C, _lib.ptr(bias), _lib.ptr(mask), -1,
                                                     nwin, (H // 7) ** 2, nH, 49, 32, int(ms[0]), int(es[0]), int(mb[0]), int(eb[0]), float(s_at),
                                                     int(mo[0]), int(eo[0]), _lib.ptr(phi), _lib.ptr(phim), st()), n=10))
            r["band"].append(timeit(lambda: _lib.call("ivit_window_attention_i8_band", _lib.ptr(qkv), _lib.ptr(out), C, _lib.ptr(bias), _lib.ptr(mask),
                                                      nwin, (H // 7) ** 2, nH, 49, 32, int(ms[0]), int(es[0]), int(mb[0]), int(eb[0]), float(s_at),
                                                      int(mo[0]), int(eo[0]), _lib.ptr(bandd), bw, band.shape[0], 0, 0, 0, 0, st()), n=10))
            r["band256"].append(timeit(lambda: _lib.call("ivit_window_attention_i8_band", _lib.ptr(qkv), _lib.ptr(out), C, _lib.ptr(bias), _lib.ptr(mask),
                                                         nwin, (H // 7) ** 2, nH, 49, 32, int(ms[0]), int(es[0]), int(mb[0]), int(eb[0]), float(s_at),
                                                         int(mo[0]), int(eo[0]), _lib.ptr(band256), bw, 256, 0, 0, 0, 0, st()), n=10))
            r["pow2"].append(timeit(lambda: _lib.call("ivit_window_attention_i8", _lib.ptr(qkv), _lib.ptr(out), C, _lib.ptr(bias), _lib.ptr(mask), -400, nwin,
                                                      (H // 7) ** 2, nH, 49, 32, int(ms[0]), int(es[0]), 1 << 30, 30, 0.25, int(mo[0]), int(eo[0]), st()), n=10))
        print(f"window attention windows={nwin:5d} heads={nH:2d} masked, natural scale s={float(s_at)}: literal {np.median(r['lit']):7.1f} us   "
              f"band table (W={bw}, {band.shape[0]} row) {np.median(r['band']):7.1f} us   256 rows through LDS {np.median(r['band256']):7.1f} us   power-of-two form {np.median(r['pow2']):7.1f} us", flush=True)
if "res" in which:
    for H, C, nH in STAGES:
        rows = B * H * H
        acc = d(rng.integers(-100000, 100000, size=(rows, C)).astype(np.int32))
        a8 = d(rng.integers(-128, 128, size=(rows, C)).astype(np.int8))
        res = d(rng.integers(-20000, 20000, size=(rows, C)).astype(np.int16))
        out = torch.empty(rows, C, dtype=torch.int16, device=DEV)
        pre = (rng.uniform(0.5, 1.0, size=C) * 2.0 ** -4).astype(np.float32)
        mp, ep = dyadic(pre, np.float32(1.0))
        mpd, epd = d(mp.view(np.int32)), d(ep)
        ma, ea = dyadic(np.float32(0.013), np.float32(0.02))
        us = timeit(lambda: _lib.call("ivit_residual_requant_i16", _lib.ptr(acc), 32, _lib.ptr(mpd), _lib.ptr(epd), int(ma[0]), int(ea[0]),
                                      _lib.ptr(res), int(ma[0]), int(ea[0]), _lib.ptr(out), rows, C, H, H, 7, 3 if H > 7 else 0, st()))
        print(f"residual i32->i16 rows={rows:7d} C={C:4d}: {us:8.1f} us  ({8 * rows * C / us / 1e3:7.1f} GB/s algorithmic)", flush=True)
        us = timeit(lambda: _lib.call("ivit_residual_requant_i16", _lib.ptr(a8), 8, None, None, int(ma[0]), int(ea[0]),
                                      _lib.ptr(res), int(ma[0]), int(ea[0]), _lib.ptr(out), rows, C, 0, 0, 0, 0, st()))
        print(f"residual i8->i16  rows={rows:7d} C={C:4d}: {us:8.1f} us  ({5 * rows * C / us / 1e3:7.1f} GB/s algorithmic)", flush=True)
