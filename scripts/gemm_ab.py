"""Interleaved A/B timing of GEMM kernel variants (debug flags): every round runs each variant once, in turn, so that clock /
thermal drift hits all variants alike; reports median and min per variant.
usage: gemm_ab.py [--resid] [--blocks] [shape ...] -- flag flag ...
A flag written a:b sets ivit_debug_set_gemm_flags(a) and ivit_debug_set_gemm_flags2(b) (cache-policy bits)."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import _lib  # noqa: E402

DEV = "cuda:0"
M = 197 * 256
SHAPES = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
RESID = "--resid" in sys.argv
QKV = "--qkv" in sys.argv         # head-major q/k/v epilogue (qkv shape only)
BLOCKS = "--blocks" in sys.argv   # both operands in the block layout (persistent kernel only)
FRAGS = "--frags" in sys.argv     # A in the block layout, W fragment-packed: the weights-in-registers kernel
BOTH = "--both" in sys.argv       # blocks and frags, interleaved
F16 = "--frags16" in sys.argv     # the 32x32x32 and the 16x16x64 form of the weights-in-registers kernel, interleaved
args = [a for a in sys.argv[1:] if a not in ("--resid", "--blocks", "--qkv", "--frags", "--both", "--frags16")]
split = args.index("--") if "--" in args else 0
names = args[:split] or list(SHAPES)
def _flag(x):
    a, _, b = x.partition(":")
    return (int(a), int(b or 0))


flags = [_flag(x) for x in args[split + 1:]] if "--" in args else [(0, 0)]
ROUNDS, ITERS = 7, 10
rng = np.random.default_rng(0)
for name in names:
    N, K = SHAPES[name]
    A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
    W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
    b = torch.zeros(N, dtype=torch.int32, device=DEV)
    m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV)
    e = torch.full((N,), 42, dtype=torch.int32, device=DEV)
    out = torch.empty(M, N, dtype=torch.int8, device=DEV)

    # variants: ("rows" | "blocks" | "frags", debug flag); --both times the block-layout persistent kernel against the
    # weights-in-registers kernel in the same process, interleaved
    At, Wt, Wf = torch.empty_like(A), torch.empty_like(W), torch.empty_like(W)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), _lib.stream_ptr())
    _lib.call("ivit_tile_operand_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wt), _lib.stream_ptr())
    _lib.call("ivit_pack_weight_frags_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), _lib.stream_ptr())
    Wf16 = torch.empty_like(W)
    _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf16), _lib.stream_ptr())
    OPS = {"rows": (A, W, 0), "blocks": (At, Wt, 3), "frags": (At, Wf, 9), "frags16": (At, Wf16, 17)}
    resid_t = torch.from_numpy(rng.integers(-128, 128, size=(M, N)).astype(np.int8)).to(DEV)

    def run(kind):
        a_, w_, LAY = OPS[kind]
        if QKV and N % 192 == 0:
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(a_), K, _lib.ptr(w_), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e),
                      _lib.ptr(out), 197, N // 192, 64, M, N, K, LAY, _lib.stream_ptr())
        elif RESID:   # the fused residual QuantAct form (attn.proj, mlp.fc2)
            _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(a_), K, _lib.ptr(w_), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e),
                      _lib.ptr(resid_t), N, 1610612736, 31, 1073741824, 32, _lib.ptr(out), N, M, N, K, LAY, _lib.stream_ptr())
        else:
            _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(a_), K, _lib.ptr(w_), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out),
                      N, M, N, K, LAY, _lib.stream_ptr())
    kinds = ["frags", "frags16"] if F16 else ["blocks", "frags"] if BOTH else ["frags"] if FRAGS else ["blocks"] if BLOCKS else ["rows"]
    if F16:      # the two forms must agree bit for bit before they are timed
        run("frags")
        ref_out = out.clone()
        out.zero_()
        run("frags16")
        torch.cuda.synchronize()
        assert torch.equal(ref_out, out), f"{name}: the 16x16x64 form differs from the 32x32x32 form"
    variants = [(k, f) for k in kinds for f in flags]
    for k in kinds:
        for _ in range(5):
            run(k)
    res = {v: [] for v in variants}
    for r in range(ROUNDS):
        for v in (variants if r % 2 == 0 else variants[::-1]):
            _lib.call("ivit_debug_set_gemm_flags", v[1][0])
            _lib.call("ivit_debug_set_gemm_flags2", v[1][1])
            run(v[0])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                run(v[0])
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / ITERS * 1e3)
    _lib.call("ivit_debug_set_gemm_flags", 0)
    _lib.call("ivit_debug_set_gemm_flags2", 0)
    for v in variants:
        t = sorted(res[v])
        med, mn = t[len(t) // 2], t[0]
        print(f"{name:5s} {v[0]:6s} flags={v[1][0]:8d}:{v[1][1]:<3d}  median {med:7.1f} us ({2 * M * N * K / med / 1e6:7.1f} TOPS)   min {mn:7.1f} us", flush=True)
