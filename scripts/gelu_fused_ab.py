"""mlp.fc1 + ShiftGELU at the headline shape (50 432 x 3072 x 768, block layouts): the GEMM followed by the stand-alone table pass
(in place) against ivit_gemm_i8_requant_gelu_ex, interleaved; HIP events around each variant, medians over rounds.
usage: gelu_fused_ab.py [batch]"""
import os
import sys
os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import _lib  # noqa: E402
from ivit_amd.prepare import dyadic  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, N, K = 197 * B, 3072, 768
g = torch.Generator(device="cpu").manual_seed(5)
A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(DEV)
W = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g).to(DEV)
b = torch.zeros(N, dtype=torch.int32, device=DEV)
m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV)
e = torch.full((N,), 44, dtype=torch.int32, device=DEV)
s_g = np.float32(0.0517)
mg, eg = dyadic(np.float32(s_g * np.float32(1 / 128)), np.float32(0.011))
lut = torch.empty(65536, dtype=torch.int8, device=DEV)
st = _lib.stream_ptr()
_lib.call("ivit_shiftgelu_build_lut_ex", float(s_g), int(mg[0]), int(eg[0]), None, _lib.ptr(lut), st)
At = torch.zeros(M * K, dtype=torch.int8, device=DEV)
_lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), st)
Wf = torch.zeros(N * K, dtype=torch.int8, device=DEV)
_lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), st)
out = torch.zeros((M + 15) * N, dtype=torch.int8, device=DEV)
ws = torch.zeros((M + 127) // 128, dtype=torch.int32, device=DEV)


def separate():
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K,
              16 | 1 | 4, st)
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(out), N, M, N, _lib.ptr(lut), _lib.ptr(out), N, 1 | 2, st)


def gemm_only():
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K,
              16 | 1 | 4, st)


def fused():
    _lib.call("ivit_gemm_i8_requant_gelu_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(lut),
              _lib.ptr(ws), _lib.ptr(out), N, M, N, K, 16 | 1 | 4, st)


def fused_nomap():
    _lib.call("ivit_debug_set_gemm_flags2", 0x1000)
    fused()
    _lib.call("ivit_debug_set_gemm_flags2", 0)


V = {"gemm only": gemm_only, "gemm + table pass": separate, "fused": fused, "fused, panels counted but not mapped": fused_nomap}
T = {k: [] for k in V}
for f in V.values():
    f()
torch.cuda.synchronize()
ref = None
for rnd in range(9):
    for k, f in V.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        T[k].append(e0.elapsed_time(e1) * 100)      # us per call
separate()
ref = out.clone()
fused()
print("fused == separate:", bool(torch.equal(ref, out)), " workspace zero:", int(ws.abs().max()) == 0)
for k, v in T.items():
    v = sorted(v)
    print(f"{k:40s} median {v[len(v) // 2]:7.1f} us   min {v[0]:7.1f} us")
