"""Reference point: the vendor library's INT8 GEMM (torch._int_mm -> hipBLASLt, int32 output, NO requant epilogue) at the
DeiT-B batch-256 shapes, random int8 data."""
import torch
M = 197 * 256
for name, (N, K) in {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}.items():
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda")
    b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
    bt = b.t()  # [K, N] column-major view, as a weight [N, K] is used
    for mat, tag in ((bt, "W[N,K]^T"), (b.t().contiguous(), "W[K,N]")):
        try:
            for _ in range(3):
                torch._int_mm(a, mat)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch._int_mm(a, mat)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"{name:5s} N={N:5d} K={K:5d} {tag:9s} {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TOPS", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(name, tag, "failed:", str(ex)[:120], flush=True)
