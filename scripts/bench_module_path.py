"""Throughput of the module-by-module path (what runs when the fused engine does not apply: I-BERT operators -- the fork's
default --, non-8-bit widths) next to the fused engine, DeiT-B.  usage: bench_module_path.py [batch] [--engine-only] [--family ibert]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd as ivit
from ivit_amd import synth
from ivit_amd.quantization_utils import QuantAct
DEV = "cuda:0"
ENGINE_ONLY = "--engine-only" in sys.argv
FAMS = [sys.argv[sys.argv.index("--family") + 1]] if "--family" in sys.argv else ["ivit", "ibert"]
_pos = [a for a in sys.argv[1:] if a.isdigit()]
B = int(_pos[0]) if _pos else 64
fs = synth.make_float_state("deit_base_patch16_224", 7)
imgs = torch.from_numpy(synth.make_images(min(B, 16), 99)).to(DEV)
imgs = imgs.repeat((B + imgs.shape[0] - 1) // imgs.shape[0], 1, 1, 1)[:B].contiguous()


HOST = {}


def timed(model, n=5):
    with torch.no_grad():
        model(imgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(imgs)
        HOST["ms"] = (time.perf_counter() - t0) / n * 1e3      # time to ISSUE the forwards (no read-back on a sync-free path)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for fam in FAMS:
    model = ivit.deit_base_patch16_224(gelu_type=fam, softmax_type=fam, layernorm_type=fam)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        model(imgs[:8])               # calibration forward (running min / max)
    ivit.freeze_model(model)
    if not ENGINE_ONLY:
        model.use_engine = False
        ms = timed(model)
        print(f"{fam:5s} module path  batch {B}: {ms:8.2f} ms / forward  ({B / ms * 1e3:8.0f} img/s), host issue time {HOST['ms']:.2f} ms", flush=True)
        if os.environ.get("IVIT_KERNEL_SPLIT"):
            from torch.profiler import profile, ProfilerActivity
            with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
                model(imgs)
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=70), flush=True)
        # the same forward captured once and replayed (nothing in it reads back): the host's launch work drops out
        with torch.no_grad():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                model(imgs)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            print(f"{fam:5s} module path  batch {B}, HIP-graph replay: {(time.perf_counter() - t0) / 10 * 1e3:8.2f} ms / forward", flush=True)
    model.use_engine = True
    print(f"{fam:5s} engine       batch {B}: {timed(model, 10):8.2f} ms / forward  (reason if not taken: {model.engine_unsupported_reason()})", flush=True)
    if os.environ.get("IVIT_KERNEL_SPLIT"):   # per-kernel time of the engine forward (torch profiler)
        from torch.profiler import profile, ProfilerActivity
        with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
            model(imgs)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=60), flush=True)
