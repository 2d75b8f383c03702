"""Probe: does splitting the batch over two HIP streams (two half-batch engines launched concurrently) beat one full-batch
forward?  (fills GEMM tail rounds / overlaps HBM-bound row kernels with MFMA-bound GEMMs, if the hardware co-schedules them)
usage: dual_stream_probe.py [nsplit]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")  # kernel-form knobs live in libivit_hip_lab.so
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import synth  # noqa: E402
from ivit_amd.checkpoint import load_synthetic_model  # noqa: E402
from ivit_amd.engine import IntViTEngine  # noqa: E402

DEV = "cuda:0"
B = 256
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
from ivit_amd import _lib  # noqa: E402
fs, ranges, cfg, meta, z = load_synthetic_model("deit_base")
imgs = torch.from_numpy(synth.make_images(16, 1003)).to(DEV).repeat(B // 16, 1, 1, 1).contiguous()


def timeit(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


full = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=B)
ref = full.forward(imgs)[0].clone()
print(f"one stream, batch {B}: {timeit(lambda: full.forward(imgs)):.3f} ms", flush=True)

engs = [IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=B // NS) for _ in range(NS)]
streams = [torch.cuda.Stream(device=DEV) for _ in range(NS)]
parts = [imgs[i * (B // NS):(i + 1) * (B // NS)].contiguous() for i in range(NS)]
outs = [None] * NS


def split():
    cur = torch.cuda.current_stream()
    for i in range(NS):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs[i] = engs[i].forward(parts[i])
    for i in range(NS):
        cur.wait_stream(streams[i])


_lib.call("ivit_debug_set_gemm_flags", FLAGS)
print(f"{NS} streams, batch {B // NS} each, gemm flags {FLAGS}: {timeit(split):.3f} ms", flush=True)
got = torch.cat([o[0] for o in outs], 0)
print("identical logits:", bool((got == ref).all()))
_lib.call("ivit_debug_set_gemm_flags", 0)
