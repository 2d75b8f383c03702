"""Wave timeline of the streaming int8 LayerNorm kernel (lab build stamps, s_memrealtime at 100 MHz -> 10 ns ticks): when do waves
start, when is the constants table ready, when are the first slots computed, when does the wave end?  Percentiles over all waves,
relative to the earliest wave's entry."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from ivit_amd.prepare import LayerNormParams
DEV = "cuda:0"
rng = np.random.default_rng(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for rows, C in ((197 * 256, 768), (197 * 256 * 4, 768)):
    x = t(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8))
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32), np.float32(2.0 ** -4))
    b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
    out = torch.empty((rows + 15) // 16 * 16, C, dtype=torch.int8, device=DEV)
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=DEV)
    for abl, cfg in ((0, 0), (1, 0), (5, 0), (0, 4)):
        _lib.call("ivit_debug_ln_ablate", abl); _lib.call("ivit_debug_ln_stream_cfg", cfg)
        call = lambda: _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), C, 1, _lib.stream_ptr())
        for _ in range(5): call()
        torch.cuda.synchronize()
        _lib.call("ivit_debug_ln_stamp_buffer", _lib.ptr(stamps))
        call(); torch.cuda.synchronize()
        _lib.call("ivit_debug_ln_stamp_buffer", None)
        st = stamps.cpu().numpy().reshape(-1, 8)
        st = st[st[:, 0] != 0]
        t0 = st[:, 0].min()
        us = lambda col: (st[:, col] - t0) / 100.0
        pct = lambda v: " ".join(f"{np.percentile(v, p):6.2f}" for p in (0, 10, 50, 90, 100))
        print(f"rows={rows} abl={abl} cfg={cfg} waves={len(st)} groups/wave {st[:, 7].min()}..{st[:, 7].max()}  [us since first wave entry: min p10 p50 p90 max]")
        for name, col in (("entry", 0), ("table ready", 1), ("slot0 done", 2), ("slot1 done", 3), ("slot2 done", 4), ("end", 6)):
            print(f"   {name:12s} {pct(us(col))}")
        print(f"   wave lifetime {pct((st[:, 6] - st[:, 0]) / 100.0)}", flush=True)
        print(f"   shader clock over the wave lifetimes (GHz) {pct(st[:, 5] / ((st[:, 6] - st[:, 0]) * 10.0))}", flush=True)
        # who is late?  mean end time by workgroup-index octile (dispatch order) and by blockIdx % 8 (XCD)
        wg = np.arange(len(st)) // 4
        oct_ = [float(us(6)[(wg * 8 // (wg.max() + 1)) == i].mean()) for i in range(8)]
        xcd = [float(us(6)[(wg % 8) == i].mean()) for i in range(8)]
        s0 = [float(us(2)[(wg % 8) == i].mean()) for i in range(8)]
        print("   mean end by workgroup octile:", " ".join(f"{v:5.1f}" for v in oct_), "| by blockIdx%8:", " ".join(f"{v:5.1f}" for v in xcd),
              "| slot0 done by blockIdx%8:", " ".join(f"{v:5.1f}" for v in s0), flush=True)
    _lib.call("ivit_debug_ln_ablate", 0); _lib.call("ivit_debug_ln_stream_cfg", 0)
