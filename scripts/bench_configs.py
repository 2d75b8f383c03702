"""Throughput of every BASELINE.json config that fits one GPU (SURVEY.md §8d "per-config throughput"):
  1 DeiT-T b1, 2 DeiT-S b64, 3 DeiT-B b256, 4 ViT-B b128 (= one rank's shard of b1024 over 8 GPUs), 5 Swin-T b128.
Synthetic weights + ranges from tests/golden, random N(0,1) images resident in HBM, whole forward timed with
events on the launch stream.  Prints one JSON line per config; `python scripts/bench_configs.py 5` runs one."""
import json
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
CONFIGS = {1: ("deit_tiny", 1), 2: ("deit_small", 64), 3: ("deit_base", 256), 4: ("vit_base", 128), 5: ("swin_tiny", 128),
           # the same models with their activation ranges AS CALIBRATED (natural scales, the regime of a real checkpoint)
           13: ("deit_base_natural", 256), 15: ("swin_tiny_natural", 128)}
GMAC = {"deit_tiny": 1.2537, "deit_small": 4.5989, "deit_base": 17.5638, "vit_base": 17.5638, "swin_tiny": 4.4906,
        "deit_base_natural": 17.5638, "swin_tiny_natural": 4.4906}


def run(cid, steps=20, warmup=5, graph=False, batch=None):
    tag, B = CONFIGS[cid]
    B = batch or B
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    if tag.startswith("swin"):
        eng = IntSwinEngine(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"], device=DEV,
                            max_batch=B)
    else:
        eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=B)
    if os.environ.get("IVIT_BLOCK_A") and hasattr(eng, "block_a"):   # A/B: e.g. IVIT_BLOCK_A=gelu or IVIT_BLOCK_A=none
        eng.block_a = {k: k in os.environ["IVIT_BLOCK_A"].split(",") for k in eng.block_a}
    if os.environ.get("IVIT_WEIGHT_FRAGS") and hasattr(eng, "weight_frags"):   # A/B: 0 = block-layout weights, LDS-DMA kernel
        eng.weight_frags = os.environ["IVIT_WEIGHT_FRAGS"] != "0"
    if os.environ.get("IVIT_PROJ_I16") and hasattr(eng, "proj_i16"):   # A/B (Swin): 0 = raw int32 accumulators out of attn.proj
        eng.proj_i16 = os.environ["IVIT_PROJ_I16"] != "0"
    if os.environ.get("IVIT_PROJ_FUSED") and hasattr(eng, "proj_fused"):   # A/B (Swin): 0 = attention in window order, proj GEMM + residual kernel
        eng.proj_fused = os.environ["IVIT_PROJ_FUSED"] != "0"
    if os.environ.get("IVIT_COMPACT_WS") and hasattr(eng, "_compact"):   # A/B: 0 = every intermediate in its own buffer
        eng._compact(os.environ["IVIT_COMPACT_WS"] != "0")
    if os.environ.get("IVIT_GELU_INPLACE") and hasattr(eng, "gelu_in_place"):   # A/B: 0 = GELU into its own buffer
        eng.gelu_in_place = os.environ["IVIT_GELU_INPLACE"] != "0"
    imgs = torch.from_numpy(synth.make_images(min(B, 16), 1000 + cid)).to(DEV)
    imgs = imgs.repeat((B + imgs.shape[0] - 1) // imgs.shape[0], 1, 1, 1)[:B].contiguous()
    if os.environ.get("IVIT_INPUT_U8") == "1" and not tag.startswith("swin"):   # uint8 pixels instead of float32 images (a quarter of the input bytes)
        imgs = torch.randint(0, 256, imgs.shape, dtype=torch.uint8, device=DEV)
    fwd = eng.forward_graph if graph else eng.forward
    for _ in range(warmup):
        fwd(imgs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fwd(imgs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    ips = B / ms * 1e3
    line = {"config": cid, "model": tag, "batch": B, "hip_graph": graph, "ms_per_forward": round(ms, 3), "images_per_s": round(ips, 1),
            "int8_tops": round(ips * GMAC[tag] * 2e9 / 1e12, 1)}
    if getattr(eng, "window_softmax_forms", None):      # natural scales: table or literal Shiftmax per attention block
        line["window_softmax_forms"] = eng.window_softmax_forms
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    # `--batches 1,8,64` : the given configs at other batch sizes (latency / throughput against the batch, e.g. for serving)
    batches = [None]
    argv = sys.argv[1:]
    if "--batches" in argv:
        i = argv.index("--batches")
        batches = [int(b) for b in argv[i + 1].split(",")]
        argv = argv[:i] + argv[i + 2:]
    args = [a for a in argv if a != "--graph"]
    for c in ([int(a) for a in args] or [k for k in sorted(CONFIGS) if k < 10]):
        for b in batches:
            run(c, batch=b)
            if "--graph" in sys.argv:
                run(c, graph=True, batch=b)
