"""A/B of the weights-in-registers GEMM's work items (16x16x64 form): 128 x 256 tiles against 128 x 128 ("narrow") tiles, forced
through the lab build (flags2 bits 14 / 13), and the launcher's own choice, at the GEMM shapes of the BASELINE configs.
Device-scope HIP events on the launch stream, interleaved rounds, best of rounds."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib, hiptime
DEV = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(1)

def time_us(fn, n=30):
    st = _lib.stream_ptr()
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = hiptime.Event(), hiptime.Event()
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_ms(e1) / n * 1e3

SHAPES = [  # (label, M, N, K, kind)
    ("DeiT-S b64 qkv", 12608, 1152, 384, "qkv6"), ("DeiT-S b64 proj", 12608, 384, 384, "res"), ("DeiT-S b64 fc1", 12608, 1536, 384, "rq"),
    ("DeiT-S b64 fc2", 12608, 384, 1536, "res"),
    ("DeiT-B b256 qkv", 50432, 2304, 768, "qkv12"), ("DeiT-B b256 proj", 50432, 768, 768, "res"), ("DeiT-B b256 fc1", 50432, 3072, 768, "rq"),
    ("DeiT-B b256 fc2", 50432, 768, 3072, "res"),
    ("ViT-B b128 qkv", 25216, 2304, 768, "qkv12"), ("ViT-B b128 proj", 25216, 768, 768, "res"), ("ViT-B b128 fc1", 25216, 3072, 768, "rq"),
    ("ViT-B b128 fc2", 25216, 768, 3072, "res"),
    ("Swin-T b128 s1 qkv", 100352, 576, 192, "rq"), ("Swin-T b128 s1 fc1", 100352, 768, 192, "rq"),
    ("Swin-T b128 s2 qkv", 25088, 1152, 384, "rq"), ("Swin-T b128 s2 fc1", 25088, 1536, 384, "rq"),
    ("DeiT-T b256 qkv", 50432, 576, 192, "qkv3"), ("DeiT-T b256 proj", 50432, 192, 192, "res"), ("DeiT-T b256 fc1", 50432, 768, 192, "rq"),
    ("DeiT-T b256 fc2", 50432, 192, 768, "res"),
]
for label, M, N, K, kind in SHAPES:
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(DEV)
    W = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g).to(DEV)
    b = torch.randint(-50000, 50000, (N,), dtype=torch.int32, generator=g).to(DEV)
    m = torch.full((N,), 1 << 30, dtype=torch.int32, device=DEV); e = torch.full((N,), 30 + 14, dtype=torch.int32, device=DEV)
    res = torch.randint(-128, 128, (M, N), dtype=torch.int8, generator=g).to(DEV)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), _lib.stream_ptr())
    Wf = torch.zeros((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), _lib.stream_ptr())
    out = torch.zeros(R16 * N, dtype=torch.int8, device=DEV)
    st = _lib.stream_ptr
    if kind == "rq":
        call = lambda: _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, 16 | 1 | 4, st())
    elif kind == "res":
        call = lambda: _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(res), N,
                                 1 << 30, 31, 1 << 30, 31, _lib.ptr(out), N, M, N, K, 16 | 1, st())
    else:
        H = int(kind[3:]); T = 197
        call = lambda: _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), T, H, 64, M, N, K, 16 | 1, st())
    res_t, outs = {}, {}
    for rnd in range(3):
        for name, f2 in (("full", 0), ("narrow", 8192), ("auto", 0)):
            _lib.call("ivit_debug_set_gemm_flags2", f2)
            out.zero_()
            res_t[name] = min(res_t.get(name, 1e9), time_us(call))
            outs[name] = out.clone()
    _lib.call("ivit_debug_set_gemm_flags2", 0)
    same = torch.equal(outs["full"], outs["narrow"])
    tops = 2.0 * M * N * K / 1e12
    print(f"{label:22s} M={M:6d} N={N:5d} K={K:5d}: full {res_t['full']:7.1f} us ({tops / res_t['full'] * 1e6:6.0f} TOPS)  narrow {res_t['narrow']:7.1f} us "
          f"({tops / res_t['narrow'] * 1e6:6.0f} TOPS)  auto {res_t['auto']:7.1f}  ratio narrow/full {res_t['narrow'] / res_t['full']:.3f}  identical {same}", flush=True)
