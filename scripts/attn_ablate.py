"""Ablation of the fused attention kernel (lab build; bits 20-24 of ivit_debug_ln_ablate): what each phase costs.
1 no score requant / clamp, 2 no table lookups, 4 no probability products, 8 no P.V + output, 16 one query tile per wave
(K / V staging + one tile: the fixed cost of a workgroup)."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd  # noqa: F401
from ivit_amd import _lib
from ivit_amd.prepare import dyadic
DEV = "cuda:0"
B, H, T, HD = 256, 12, 197, 64
rng = np.random.default_rng(0)
qkv = torch.from_numpy(np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, HD))), -128, 127).astype(np.int8)).to(DEV)
out = torch.empty(B * T, H * HD, dtype=torch.int8, device=DEV)
ms, es = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -2))
mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))


def run():
    _lib.call("ivit_attention_fused_i8", _lib.ptr(qkv), _lib.ptr(out), B, H, T, HD, int(ms[0]), int(es[0]), 0.25, int(mo[0]), int(eo[0]),
              _lib.stream_ptr())


# bit 6 (= bit 26 of the knob): the float64 requantisation of the scores instead of the float32 one (A/B of attention_kernel<.., RQ32>)
for bits in [int(x) for x in (sys.argv[1:] or ["0", "64", "1", "2", "4", "8", "3", "7", "15", "16", "31"])]:
    _lib.call("ivit_debug_ln_ablate", bits << 20)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"ablate {bits:2d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
_lib.call("ivit_debug_ln_ablate", 0)
