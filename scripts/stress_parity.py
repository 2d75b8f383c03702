"""Run the fused engines many times on the same resident batch and count forwards whose INT32 logits differ from the first one:
a rare hazard in the hand-scheduled kernels (asynchronous loads behind inline asm) would show as a non-zero count.
usage: stress_parity.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
from ivit_amd import synth  # noqa: E402
from ivit_amd.checkpoint import load_synthetic_model  # noqa: E402
from ivit_amd.engine import IntViTEngine  # noqa: E402
from ivit_amd.swin_engine import IntSwinEngine  # noqa: E402

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for tag, B in (("deit_base", 256), ("deit_base_natural", 256), ("vit_base", 128), ("deit_small", 64), ("swin_tiny", 128), ("swin_tiny_natural", 128)):
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    if tag.startswith("swin"):
        eng = IntSwinEngine(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"], device=DEV, max_batch=B)
    else:
        eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=B)
    imgs = torch.from_numpy(synth.make_images(16, 77)).to(DEV).repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
    ref = eng.forward(imgs)[0].clone()
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    for it in range(N):
        out = (eng.forward_graph if it % 2 else eng.forward)(imgs)[0]
        bad += (out != ref).any().to(torch.int64)
    torch.cuda.synchronize()
    print(f"{tag:20s} batch {B}: {N} forwards, {int(bad)} differ from the first", flush=True)
    del eng
    torch.cuda.empty_cache()
