"""Deterministic synthetic weights / images for the integer-only ViT path.

There is no network on the build or GPU boxes, so every benchmark and parity
case runs on random-init weights of the named architecture.
The generator is counter based (numpy Philox keyed by ``(seed, crc32(name))``)
so a tensor's values do not depend on generation order and are bit-identical
in this container and on the GPU box (same numpy build).  The golden fixtures
under ``tests/golden/`` store SHA-256 digests of the derived int8 weights to
detect any drift.

Distributions follow the reference initialisation
(/root/reference/models/vit_quant.py:272-283, layers_quant.py:25-81):
Linear / Conv weights, ``cls_token`` and ``pos_embed`` are N(0, 0.02^2)
truncated to +-2 sigma.  Deviations, all deliberate:

* ``attn.qkv.weight`` is multiplied by ``QKV_GAIN`` (10) - with the stock
  init every Shiftmax output is 0 and the attention path is never exercised;
* LayerNorm gamma ~ U(0.5, 1.5), beta ~ N(0, 0.1^2) instead of (1, 0);
* biases ~ N(0, 0.02^2) instead of 0 so the int32 bias path is exercised.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

QKV_GAIN = 10.0

# name -> (embed_dim, depth, num_heads); patch 16, 224x224, mlp_ratio 4, 1000 classes
# (/root/reference/models/vit_quant.py:315-388)
MODEL_CONFIGS = {
    "deit_tiny_patch16_224": dict(embed_dim=192, depth=12, num_heads=3),
    "deit_small_patch16_224": dict(embed_dim=384, depth=12, num_heads=6),
    "deit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_large_patch16_224": dict(embed_dim=1024, depth=24, num_heads=16),
}

IMG_SIZE = 224
PATCH = 16
NUM_PATCHES = (IMG_SIZE // PATCH) ** 2
NUM_TOKENS = NUM_PATCHES + 1
NUM_CLASSES = 1000


def _rng(seed: int, name: str) -> np.random.Generator:
    key = np.array([np.uint64(seed), np.uint64(zlib.crc32(name.encode()))], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def _trunc_normal(seed, name, shape, std=0.02):
    g = _rng(seed, name)
    x = g.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    # replace the tails by a uniform draw inside +-2 sigma (deterministic, single pass)
    u = g.uniform(-2.0, 2.0, size=shape)
    x = np.where(bad, u, x)
    return (x * std).astype(np.float32)


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(size=shape) * std).astype(np.float32)


def _uniform(seed, name, shape, lo, hi):
    return _rng(seed, name).uniform(lo, hi, size=shape).astype(np.float32)


def make_float_state(model: str, seed: int = 0, depth: int | None = None) -> "OrderedDict[str, np.ndarray]":
    """Float32 parameters with the reference's state_dict names."""
    cfg = MODEL_CONFIGS[model]
    C = cfg["embed_dim"]
    D = cfg["depth"] if depth is None else depth
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(prefix, out_f, in_f, gain=1.0):
        sd[prefix + ".weight"] = _trunc_normal(seed, prefix + ".weight", (out_f, in_f)) * np.float32(gain)
        sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (out_f,), 0.02)

    def ln(prefix, n):
        sd[prefix + ".weight"] = _uniform(seed, prefix + ".weight", (n,), 0.5, 1.5)
        sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n,), 0.1)

    sd["cls_token"] = _trunc_normal(seed, "cls_token", (1, 1, C))
    sd["pos_embed"] = _trunc_normal(seed, "pos_embed", (1, NUM_TOKENS, C))
    sd["patch_embed.proj.weight"] = _trunc_normal(seed, "patch_embed.proj.weight", (C, 3, PATCH, PATCH))
    sd["patch_embed.proj.bias"] = _normal(seed, "patch_embed.proj.bias", (C,), 0.02)
    for i in range(D):
        p = f"blocks.{i}."
        ln(p + "norm1", C)
        lin(p + "attn.qkv", 3 * C, C, gain=QKV_GAIN)
        lin(p + "attn.proj", C, C)
        ln(p + "norm2", C)
        lin(p + "mlp.fc1", 4 * C, C)
        lin(p + "mlp.fc2", C, 4 * C)
    ln("norm", C)
    lin("head", NUM_CLASSES, C)
    return sd


def make_images(batch: int, seed: int, start: int = 0) -> np.ndarray:
    """float32 N(0,1) images [batch,3,224,224]; image i depends only on (seed, start+i)."""
    out = np.empty((batch, 3, IMG_SIZE, IMG_SIZE), dtype=np.float32)
    for i in range(batch):
        out[i] = _rng(seed, f"image.{start + i}").standard_normal(size=(3, IMG_SIZE, IMG_SIZE)).astype(np.float32)
    return out


def qact_names(depth: int = 12, family: str = "ivit"):
    """Names of every QuantAct in a DeiT/ViT model, in module order
    (/root/reference/models/vit_quant.py:201-248, layers_quant.py:129-188).  With the I-BERT operators the softmax
    module owns one more (16 bit, ibert_modules.py:260), registered after attn.qact3 (vit_quant.py:41-59)."""
    names = ["qact_input", "patch_embed.qact", "qact_pos", "qact1"]
    for i in range(depth):
        p = f"blocks.{i}."
        names += [p + "qact1", p + "attn.qact1", p + "attn.qact_attn1", p + "attn.qact2", p + "attn.qact3"]
        if family == "ibert":
            names += [p + "attn.int_softmax.act"]
        names += [p + "qact2", p + "qact3", p + "mlp.qact_gelu", p + "mlp.qact1", p + "mlp.qact2", p + "qact4"]
    names += ["qact2"]
    return names


# ------------------------------------------------------------------------------------------------ Swin
# (/root/reference/models/swin_quant.py:567-585): patch 4, window 7, embed 96, depths (2,2,6,2), heads (3,6,12,24)
SWIN_CONFIGS = {
    "swin_tiny_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=7),
    "swin_small_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window=7),
    "swin_base_patch4_window7_224": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=7),
    # Swin-B widths with two blocks per stage: a test-sized stand-in that exercises C = 128 * 2^k and heads 4..32
    "swin_base_shallow": dict(embed_dim=128, depths=(2, 2, 2, 2), num_heads=(4, 8, 16, 32), window=7),
}
REL_POS_STD = 0.5   # reference init is 0.02 (swin_quant.py:112); widened so the bias path is exercised


def make_swin_float_state(model: str, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Float32 parameters with the reference's Swin state_dict names."""
    cfg = SWIN_CONFIGS[model]
    C0, depths, heads, ws = cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"]
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(prefix, out_f, in_f, gain=1.0, bias=True):
        sd[prefix + ".weight"] = _trunc_normal(seed, prefix + ".weight", (out_f, in_f)) * np.float32(gain)
        if bias:
            sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (out_f,), 0.02)

    def ln(prefix, n):
        sd[prefix + ".weight"] = _uniform(seed, prefix + ".weight", (n,), 0.5, 1.5)
        sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n,), 0.1)

    sd["patch_embed.proj.weight"] = _trunc_normal(seed, "patch_embed.proj.weight", (C0, 3, 4, 4), std=0.1)
    sd["patch_embed.proj.bias"] = _normal(seed, "patch_embed.proj.bias", (C0,), 0.02)
    ln("patch_embed.norm", C0)
    for li, (depth, nh) in enumerate(zip(depths, heads)):
        C = C0 * 2 ** li
        for bi in range(depth):
            p = f"layers.{li}.blocks.{bi}."
            ln(p + "norm1", C)
            sd[p + "attn.relative_position_bias_table"] = _normal(seed, p + "attn.relative_position_bias_table",
                                                                  ((2 * ws - 1) ** 2, nh), REL_POS_STD)
            lin(p + "attn.qkv", 3 * C, C, gain=QKV_GAIN)
            lin(p + "attn.proj", C, C)
            ln(p + "norm2", C)
            lin(p + "mlp.fc1", 4 * C, C)
            lin(p + "mlp.fc2", C, 4 * C)
        if li < len(depths) - 1:
            p = f"layers.{li}.downsample."
            ln(p + "norm", 4 * C)
            lin(p + "reduction", 2 * C, 4 * C, bias=False)
    Cl = C0 * 2 ** (len(depths) - 1)
    ln("norm", Cl)
    lin("head", NUM_CLASSES, Cl)
    return sd


def swin_qact_names(depths=(2, 2, 6, 2)):
    """Every QuantAct of the reference's SwinTransformer that its forward calls, in module order
    (swin_quant.py:459-518; `act_out` is constructed but never called, :518,563)."""
    names = ["qact_input", "patch_embed.qact_before_norm", "patch_embed.qact", "qact1"]
    for li, depth in enumerate(depths):
        for bi in range(depth):
            p = f"layers.{li}.blocks.{bi}."
            names += [p + "qact1", p + "attn.qact1", p + "attn.qact_attn1", p + "attn.qact_table", p + "attn.qact2",
                      p + "attn.qact3", p + "attn.qact4", p + "qact2", p + "qact3", p + "mlp.qact_gelu", p + "mlp.qact1",
                      p + "mlp.qact2", p + "qact4"]
        if li < len(depths) - 1:
            names += [f"layers.{li}.downsample.qact1", f"layers.{li}.downsample.qact2"]
    names += ["qact2", "qact3"]
    return names
