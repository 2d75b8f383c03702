"""DeiT-B b256 forward (lab library) with the attention kernel at 4 and at 3 workgroups per CU, interleaved."""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ivit_amd  # noqa: F401
from ivit_amd import _lib, synth
from ivit_amd.checkpoint import load_synthetic_model
from ivit_amd.engine import IntViTEngine
DEV = "cuda:0"
B = 256
fs, ranges, cfg, meta, z = load_synthetic_model("deit_base")
eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=B)
imgs = torch.from_numpy(synth.make_images(16, 1003)).to(DEV).repeat(B // 16, 1, 1, 1).contiguous()


def timeit(steps=20, warmup=5):
    for _ in range(warmup):
        eng.forward(imgs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.forward(imgs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for rnd in range(4):
    for name, bits in (("4 workgroups / CU", 0), ("3 workgroups / CU", 1 << 25)):
        _lib.call("ivit_debug_ln_ablate", bits)
        print(f"round {rnd}  attention at {name}: {timeit():.3f} ms / forward", flush=True)
_lib.call("ivit_debug_ln_ablate", 0)
