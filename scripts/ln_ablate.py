"""Where does the time of the int8 LayerNorm go?  Times the streaming kernel (ln_stream.h) and the round-3 grouped kernel at the
headline shape with parts removed (lab build, ivit_debug_ln_ablate; results WRONG when a bit is set): 1 no element chain,
2 no statistics (Newton steps), 4 no stores, 8 no table build (grouped kernel only)."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib, hiptime
from ivit_amd.prepare import LayerNormParams
DEV = "cuda:0"
rng = np.random.default_rng(0)
def time_us(fn, n=40):
    st = _lib.stream_ptr()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = hiptime.Event(), hiptime.Event()
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3
for rows, C in ((197 * 256, 768), (197 * 256 * 4, 768)):
    x = torch.from_numpy(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8)).to(DEV)
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32), np.float32(2.0 ** -4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
    out = torch.empty((rows + 15) // 16 * 16, C, dtype=torch.int8, device=DEV)
    for form, name in ((0, "stream"), (3, "grouped")):
        res = {}
        _lib.call("ivit_debug_ln_wave_per_row", form)
        for bits in (0, 1, 2, 3, 4, 5, 7):
            _lib.call("ivit_debug_ln_ablate", bits)
            res[bits] = round(time_us(lambda: _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), C, 1, _lib.stream_ptr())), 1)
        _lib.call("ivit_debug_ln_ablate", 0)
        print(f"rows={rows} C={C} {name}: us by ablation bits (1 chain, 2 stats, 4 stores) {res}", flush=True)
_lib.call("ivit_debug_ln_wave_per_row", 0)
