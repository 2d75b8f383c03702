"""One int8 LayerNorm form, N launches at one shape (for rocprofv3 --pmc passes): python scripts/ln_one.py FORM CFG [ROWS] [COMPAT]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from ivit_amd.prepare import LayerNormParams, phi_tables
form, cfg = int(sys.argv[1]), int(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 197 * 256
compat = int(sys.argv[4]) if len(sys.argv) > 4 else 0
C, DEV = 768, "cuda:0"
rng = np.random.default_rng(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
x = t(np.clip(np.rint(rng.normal(0, 30, size=(rows, C))), -128, 127).astype(np.int8))
lp = LayerNormParams(rng.uniform(0.5, 1.5, size=C).astype(np.float32), rng.normal(0, 0.1, size=C).astype(np.float32), np.float32(2.0 ** -4))
b, s, m, e = t(lp.bias_int), t(lp.s_ln), t(lp.m.view(np.int32)), t(lp.e)
remap, phi = phi_tables(np.float32(0.0371)); remap, phi = t(remap), t(phi)
out = torch.empty((rows + 15) // 16 * 16, C, dtype=torch.int8, device=DEV)
_lib.call("ivit_debug_ln_wave_per_row", form); _lib.call("ivit_debug_ln_stream_cfg", cfg)
for _ in range(20):
    if compat:
        _lib.call("ivit_layernorm_i8_compat", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e), _lib.ptr(remap), _lib.ptr(phi), _lib.ptr(out), C, 1, _lib.stream_ptr())
    else:
        _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x), C, rows, C, _lib.ptr(b), _lib.ptr(s), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), C, 1, _lib.stream_ptr())
torch.cuda.synchronize()
