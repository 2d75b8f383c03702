"""Checkpoint loading, calibration warm-up and dataset evaluation with the reference's harness semantics
(/root/reference/scripts/inference.py:33-91 calibrate_model, :94-224 load_model, :231-267 evaluate_dataset;
checkpoint layout written by /root/reference/quant_train.py:470-500), on the MI355X integer path.

What is the same: the checkpoint dict (`'model'` state_dict + optional `'model_config'`, or a bare state_dict), the
override arguments, the scalar -> [1] buffer fix-up, `strict_load` skipping the warm-up, `freeze_model` at the end,
and the (top-1, top-3, top-5) percentages `evaluate_dataset` returns.

What differs, deliberately:
  * the reference always instantiates `deit_tiny_patch16_224` whatever `model_config['model_name']` says (:133); here
    the name selects the factory when it is one this package provides (DeiT-T/S/B, ViT-B/L, Swin-T/S/B) and falls back
    to DeiT-T otherwise;
  * operator families: 'ivit' (fused engine) and 'ibert' (module path) exist; the reference's DEFAULTS are kept -- 'ibert'
    when the saved configuration lacks a type (:111-113), and for a checkpoint without any configuration the reference
    picks its 'ppoly_...' GELU / softmax (:160-162), which this path does not implement: that case raises with a message
    naming the override to pass, instead of silently substituting another operator;
  * calibration takes any iterable of image batches (no torchvision / ImageNet reader in this environment); the
    single random warm-up forward of `use_random_calibration` is kept as is.
"""
from __future__ import annotations

import time
from typing import Iterable, Optional

import torch

from . import swin_quant, vit_quant
from .model_utils import freeze_model

FACTORIES = {name: getattr(mod, name) for mod in (vit_quant, swin_quant) for name in mod.__all__ if name.endswith("_224")}
_BW_KEYS = ("patch_embed_bw", "pos_encoding_bw", "block_input_bw", "attention_out_bw", "softmax_bw", "mlp_out_bw",
            "norm2_in_bw", "att_block_out_bw")


def calibrate_model(model, device, batches: Optional[Iterable[torch.Tensor]] = None, use_random_calibration: bool = False):
    """Running-stat forward passes that initialise / update every QuantAct range (inference.py:33-91)."""
    model.eval()
    with torch.no_grad():
        if use_random_calibration:
            model(torch.randn(1, 3, 224, 224, device=device))
            return
        if batches is None:
            raise ValueError("batches is required when use_random_calibration=False")
        for imgs in batches:
            if isinstance(imgs, (tuple, list)):   # (images, targets) pairs of a data loader
                imgs = imgs[0]
            model(imgs.to(device))


def build_model(model_config: Optional[dict] = None, num_classes: int = 1000, gelu_type=None, softmax_type=None,
                layernorm_type=None, bitwidth=None):
    """The model construction half of load_model (inference.py:100-189)."""
    cfg = dict(model_config or {})
    name = cfg.get("model_name", "deit_tiny")
    factory = FACTORIES.get(name) or FACTORIES.get(f"{name}_patch16_224") or FACTORIES["deit_tiny_patch16_224"]
    if model_config is None:
        # inference.py:160-162: a configuration-less checkpoint gets the reference's ppoly GELU / softmax and I-BERT LayerNorm
        if gelu_type is None or softmax_type is None:
            raise KeyError("checkpoint without 'model_config': the reference would build its 'ppoly_deg_2_seg_16_...' GELU / "
                           "softmax here (scripts/inference.py:160-162), which the MI355X integer path does not implement -- "
                           "pass gelu_type= / softmax_type= ('ivit' or 'ibert') explicitly")
        ops = dict(gelu_type=gelu_type, softmax_type=softmax_type,
                   layernorm_type=layernorm_type if layernorm_type is not None else "ibert")
    else:
        ops = dict(gelu_type=gelu_type if gelu_type is not None else cfg.get("gelu_type", "ibert"),       # :111-113
                   softmax_type=softmax_type if softmax_type is not None else cfg.get("softmax_type", "ibert"),
                   layernorm_type=layernorm_type if layernorm_type is not None else cfg.get("layernorm_type", "ibert"))
    if factory.__name__.startswith("swin"):
        for k, v in ops.items():
            if not str(v).lower().startswith("ivit"):
                raise KeyError(f"{k}={v!r}: the Swin models of the MI355X integer path implement the 'ivit' operators only (DeiT / ViT: 'ivit' and 'ibert')")
        return factory(pretrained=False, num_classes=cfg.get("num_classes", num_classes),
                       drop_rate=cfg.get("drop_rate", 0.0), drop_path_rate=cfg.get("drop_path_rate", 0.1))
    bws = {k: (bitwidth if bitwidth is not None else cfg.get(k, 8)) for k in _BW_KEYS}
    return factory(pretrained=False, num_classes=cfg.get("num_classes", num_classes), drop_rate=cfg.get("drop_rate", 0.0),
                   drop_path_rate=cfg.get("drop_path_rate", 0.1), **bws, **ops)


def load_model(checkpoint_path, device="cuda", num_classes=1000, gelu_type=None, softmax_type=None, layernorm_type=None,
               bitwidth=None, calibration_batches: Optional[Iterable[torch.Tensor]] = None, strict_load=False,
               use_random_calibration_warmup=False):
    """inference.py:94-224: build from the saved configuration (arguments override it), load the weights, warm up the
    quantisation ranges unless `strict_load`, freeze."""
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    has_cfg = isinstance(checkpoint, dict) and "model_config" in checkpoint
    model = build_model(checkpoint["model_config"] if has_cfg else None, num_classes, gelu_type, softmax_type,
                        layernorm_type, bitwidth)
    weights = checkpoint["model"] if isinstance(checkpoint, dict) and "model" in checkpoint else checkpoint
    weights = dict(weights)
    own = model.state_dict()
    for name, param in weights.items():
        if name in own and param.shape == torch.Size([]) and own[name].shape == torch.Size([1]):
            weights[name] = param.unsqueeze(0)   # scalar buffers of older checkpoints (:203-207)
    model.load_state_dict(weights, strict=strict_load)
    model.to(device)
    if not strict_load:
        if use_random_calibration_warmup:
            calibrate_model(model, device, use_random_calibration=True)
        elif calibration_batches is not None:
            calibrate_model(model, device, calibration_batches)
    freeze_model(model)
    return model


def save_checkpoint(model, path, model_config: Optional[dict] = None, **extra):
    """The inference-relevant part of the dict quant_train.py:487-497 saves."""
    d = {"model": model.state_dict()}
    if model_config is not None:
        d["model_config"] = dict(model_config)
    d.update(extra)
    torch.save(d, path)


def evaluate_dataset(model, data_loader, device, *, print_batch_stats: bool = True):
    """inference.py:231-267 -> (top-1, top-3, top-5) accuracy in percent over (images, targets) batches."""
    correct1 = correct3 = correct5 = tot = 0
    batch_times = []
    start_total = time.perf_counter()
    model.eval()
    with torch.no_grad():
        for imgs, targets in data_loader:
            t0 = time.perf_counter()
            imgs, targets = imgs.to(device), targets.to(device)
            logits = model(imgs)
            pred5 = logits.topk(5, dim=1).indices
            hit = pred5 == targets.reshape(-1, 1)
            correct1 += int(hit[:, 0].sum())
            correct3 += int(hit[:, :3].any(dim=1).sum())
            correct5 += int(hit.any(dim=1).sum())
            tot += imgs.size(0)
            batch_times.append(time.perf_counter() - t0)
    total_time = time.perf_counter() - start_total
    if print_batch_stats and batch_times:
        print(f"Finished evaluation: total={total_time:.1f}s | avg/batch={sum(batch_times) / len(batch_times) * 1000:.1f} ms | "
              f"avg/img={(total_time / tot if tot else 0.0) * 1000:.2f} ms")
    if tot == 0:
        return 0.0, 0.0, 0.0
    return 100 * correct1 / tot, 100 * correct3 / tot, 100 * correct5 / tot
