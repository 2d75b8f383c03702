"""Composite blocks of the quantised ViT (/root/reference/models/layers_quant.py:105-203): Mlp, PatchEmbed and the
eval-mode identities (DropPath).  Module names and order define the checkpoint keys, so they follow the
reference; every operator inside is the HIP-backed one from quantization_utils."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .quantization_utils import QuantAct, QuantConv2d, QuantLinear


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """In-place truncated normal by inverse-CDF sampling (same distribution as layers_quant.py:25-81)."""
    with torch.no_grad():
        lo = 0.5 * (1.0 + math.erf((a - mean) / std / math.sqrt(2.0)))
        hi = 0.5 * (1.0 + math.erf((b - mean) / std / math.sqrt(2.0)))
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return tensor


class DropPath(nn.Module):
    """Stochastic depth; identity in eval mode, which is the only mode the integer path runs in."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.training or not self.drop_prob:
            return x
        keep = 1.0 - self.drop_prob
        mask = (torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), device=x.device) < keep).to(x.dtype)
        return x / keep * mask


class Mlp(nn.Module):
    """fc1 -> qact_gelu -> act -> qact1 -> fc2 -> qact2 (layers_quant.py:116-154)."""

    def __init__(self, in_features, act_layer, hidden_features=None, out_features=None, drop=0.0, bitwidth_out=8):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = QuantLinear(in_features, hidden_features)
        self.qact_gelu = QuantAct()
        self.act = act_layer()
        self.qact1 = QuantAct()
        self.fc2 = QuantLinear(hidden_features, out_features)
        self.qact2 = QuantAct(bitwidth_out)
        self.drop = nn.Dropout(drop)

    def forward(self, x, act_scaling_factor):
        x, s = self.fc1(x, act_scaling_factor)
        x, s = self.qact_gelu(x, s)
        x, s = self.act(x, s)
        x, s = self.qact1(x, s)
        x, s = self.fc2(self.drop(x), s)
        x, s = self.qact2(x, s)
        return self.drop(x), s


class PatchEmbed(nn.Module):
    """Strided-convolution patch embedding (layers_quant.py:157-203)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, bitwidth_out=8):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.norm_layer = norm_layer
        self.proj = QuantConv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        if norm_layer:
            self.qact_before_norm = QuantAct()
            self.norm = norm_layer(embed_dim)
        self.qact = QuantAct(bitwidth_out)

    def forward(self, x, act_scaling_factor):
        B, C, H, W = x.shape
        assert (H, W) == self.img_size, f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        x, s = self.proj(x, act_scaling_factor)
        x = x.flatten(2).transpose(1, 2)
        if self.norm_layer:
            x, s = self.qact_before_norm(x, s)
            x, s = self.norm(x, s)
        return self.qact(x, s)
