"""Calibration ranges ("checkpoint minus weights") for the synthetic models.

A frozen I-ViT model is its float parameters plus one (x_min, x_max) pair per QuantAct
(/root/reference/models/quantization_utils/quant_modules.py:264-266, 290-294).  With no
network there are no trained checkpoints, so the ranges that go with the synthetic weights
(i-vit_amd/synth.py) were produced by running the reference's own calibration forward in the
build container and snapping to +-127*2^p (oracle/gen_golden.py); they are stored next to the
golden vectors under tests/golden/<tag>.npz.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

TAGS = {"deit_tiny": "deit_tiny_patch16_224", "deit_small": "deit_small_patch16_224",
        "deit_base": "deit_base_patch16_224", "vit_base": "vit_base_patch16_224",
        "swin_tiny": "swin_tiny_patch4_window7_224", "deit_tiny_ibert": "deit_tiny_patch16_224",
        "deit_tiny_natural": "deit_tiny_patch16_224", "deit_tiny_w16": "deit_tiny_patch16_224", "deit_tiny_w16all": "deit_tiny_patch16_224", "deit_tiny_ibert_w16all": "deit_tiny_patch16_224", "swin_tiny_natural": "swin_tiny_patch4_window7_224", "deit_small_natural": "deit_small_patch16_224",
        "deit_base_natural": "deit_base_patch16_224", "deit_tiny_ibert_natural": "deit_tiny_patch16_224",
        "vit_large": "vit_large_patch16_224", "vit_large_natural": "vit_large_patch16_224",
        "swin_small": "swin_small_patch4_window7_224", "swin_small_natural": "swin_small_patch4_window7_224"}


def load_fixture(tag: str):
    z = np.load(os.path.join(GOLDEN_DIR, f"{tag}.npz"))
    meta = json.loads(str(z["meta"]))
    ranges = {str(n): (np.float32(lo), np.float32(hi)) for n, lo, hi in zip(z["range_names"], z["x_min"], z["x_max"])}
    return z, meta, ranges


def load_synthetic_model(tag: str):
    """-> (float_state, ranges, cfg, meta, fixture) for one of the committed synthetic models."""
    z, meta, ranges = load_fixture(tag)
    if meta["factory"] in synth.SWIN_CONFIGS:
        cfg = synth.SWIN_CONFIGS[meta["factory"]]
        fs = synth.make_swin_float_state(meta["factory"], meta["weight_seed"])
    else:
        cfg = synth.MODEL_CONFIGS[meta["factory"]]
        fs = synth.make_float_state(meta["factory"], meta["weight_seed"])
    return fs, ranges, cfg, meta, z
