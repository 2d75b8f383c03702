#!/usr/bin/env python3
"""Headline benchmark: images/s of the integer-only DeiT-B forward (batch 256 per GPU, 224x224
synthetic images resident in HBM) on N MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward of the whole hot path over one batch: input quantisation + im2col, patch-embed
GEMM, 12 blocks (LN, qkv GEMM, fused attention, proj GEMM + residual, LN, fc1 GEMM, ShiftGELU,
fc2 GEMM + residual), final LN, head GEMM, per-class scaling + arg-max; for N > 1 followed by the one
RCCL all-gather of the top-1 indices.  Images are independent, so ranks share nothing else (weak
scaling: 256 images per GPU).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL_TAG = "deit_base"
BATCH = 256
MAC_PER_IMAGE = 17.5638e9          # SURVEY.md Appendix C (GEMM + attention + patch-embed + head)
INT8_PEAK_TOPS = 5033.0            # 256 CU x 4 SIMD x 1024 MAC/clk x 2.4 GHz x 2 ops (MI355X_MICROARCH.md)


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE
    passes, corrected as MI355X_MICROARCH.md prescribes; scripts/summarize_profiles.py).  None if no summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d["pmc"][kernel_key]["hbm_bytes_per_launch_corrected"], os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


def cpu_baseline(fs, ranges, cfg):
    """The CPU oracle (port of the reference's integer algorithm, oracle/ivit_oracle.c) timed on this
    host's cores on a bounded sample of the same workload."""
    import numpy as np
    from ivit_amd import synth
    from oracle import oracle as orc
    n = 16
    imgs = synth.make_images(n, 31337)
    om = orc.OracleViT(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"])
    t0 = time.perf_counter()
    om.forward(imgs)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": orc.max_threads(), "kind": "port",
            "sample": f"DeiT-B INT8, one forward of {n} images (224x224 synthetic), {dt:.1f} s, "
                      f"OpenMP threads={orc.max_threads()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from ivit_amd import synth
    from ivit_amd.checkpoint import load_synthetic_model
    from ivit_amd.engine import IntViTEngine
    from ivit_amd.parallel import DataParallelTop1

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    fs, ranges, cfg, meta, _ = load_synthetic_model(MODEL_TAG)
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=dev, max_batch=BATCH)
    dp = DataParallelTop1(eng, world)
    images = torch.from_numpy(synth.make_images(BATCH, 5000 + rank)).to(dev)  # resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dp.step(images)
    sync()
    probe = []
    eng.probe = probe
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dp.step(images)
    sync()
    dt = time.perf_counter() - t0
    eng.probe = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * BATCH * args.steps / dt
        # dominant kernel: gemm_i8_pers_kernel<EPI_RESID> (attn.proj and mlp.fc2 with the fused residual QuantAct)
        ms = [e0.elapsed_time(e1) for e0, e1, *_ in probe]
        macs = [float(M) * N * K for _, _, M, N, K in probe]
        avg_ms = sum(ms) / len(ms)
        achieved = 2.0 * sum(macs) / (sum(ms) * 1e-3) / 1e12
        traffic, traffic_src = None, None
        for key in ("gemm_i8_pers_kernel<1, 0>", "gemm_i8_pers_kernel<1>", "gemm_i8_big_kernel<1, 0>"):   # name as profiled
            if traffic is None:
                traffic, traffic_src = pmc_traffic(key)
        roof = {"bound": "mfma", "kernel": "gemm_i8_pers_kernel<EPI_RESID> (attn.proj + mlp.fc2, residual QuantAct fused)",
                "achieved": round(achieved, 1), "peak": INT8_PEAK_TOPS, "unit": "TFLOP/s",
                "frac": round(achieved / INT8_PEAK_TOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches": len(ms), "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_ops_per_launch": 2.0 * sum(macs) / len(macs)}
        out = {"metric": "images/sec DeiT-B INT8 @batch256, 1→8 MI355X; % INT8 MFMA peak",
               "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
               "config": {"workload": "DeiT-B INT8 integer-only forward, batch 256 per GPU, 224x224",
                          "global_batch": world * BATCH, "parallelism": f"dp{world}"},
               "mfma_util_end_to_end": round(value / world * MAC_PER_IMAGE / 2.5166e15, 4),
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fs, ranges, cfg)
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
