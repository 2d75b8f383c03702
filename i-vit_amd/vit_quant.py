"""DeiT / ViT assembly with the reference's module tree and checkpoint keys
(/root/reference/models/vit_quant.py:23-406), running on the MI355X integer kernels.

Two execution paths give bit-identical results:
  * a frozen model (every QuantAct fixed, `freeze_model`) forwards through the fused int8 engine
    (engine.IntViTEngine): int8 activations end to end, ~100 kernel launches per batch;
  * otherwise (calibration, debugging, module-level tests) the modules are called one by one exactly as the
    reference's forward does, each converting float <-> integer views around its HIP kernel.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from .dispatch import EngineDispatch
from .layers_quant import DropPath, Mlp, PatchEmbed, trunc_normal_
from .quantization_utils import (QuantAct, QuantLinear, QuantMatMul, get_gelu, get_layernorm, get_softmax)
from .quantization_utils import lazy

__all__ = ["deit_tiny_patch16_224", "deit_small_patch16_224", "deit_base_patch16_224", "vit_base_patch16_224",
           "vit_large_patch16_224", "VisionTransformer"]


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, bitwidth_out=8,
                 bitwidth_softmax=8, softmax_cls=nn.Softmax):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = QuantLinear(dim, dim * 3, bias=qkv_bias)
        self.qact1 = QuantAct()
        self.qact_attn1 = QuantAct()
        self.qact2 = QuantAct()
        self.proj = QuantLinear(dim, dim)
        self.qact3 = QuantAct(bitwidth_out)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.int_softmax = softmax_cls(bitwidth_softmax)
        self.matmul_1 = QuantMatMul()
        self.matmul_2 = QuantMatMul()

    def forward(self, x, act_scaling_factor):
        B, N, C = x.shape
        x, s = self.qkv(x, act_scaling_factor)
        x, s_qkv = self.qact1(x, s)
        q, k, v = x.reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
        attn, s = self.matmul_1(q, s_qkv, k.transpose(-2, -1), s_qkv)
        attn, s = self.qact_attn1(attn * self.scale, s * self.scale)
        attn, s = self.int_softmax(attn, s)
        x, s = self.matmul_2(self.attn_drop(attn), s, v, s_qkv)
        x, s = self.qact2(x.transpose(1, 2).reshape(B, N, C), s)
        x, s = self.proj(x, s)
        x, s = self.qact3(x, s)
        return self.proj_drop(x), s


class Block(nn.Module):
    def __init__(self, dim, num_heads, softmax_cls, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attention_out_bw=8,
                 softmax_bw=8, mlp_out_bw=8, norm2_in_bw=8, att_block_out_bw=8):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.qact1 = QuantAct()
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, bitwidth_out=attention_out_bw, bitwidth_softmax=softmax_bw,
                              softmax_cls=softmax_cls)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.qact2 = QuantAct(norm2_in_bw)
        self.norm2 = norm_layer(dim)
        self.qact3 = QuantAct()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop,
                       bitwidth_out=mlp_out_bw)
        self.qact4 = QuantAct(att_block_out_bw)

    def forward(self, x_1, s_1):
        x, s = self.norm1(x_1, s_1)
        x, s = self.qact1(x, s)
        x, s = self.attn(x, s)
        x_2, s_2 = self.qact2(self.drop_path(x), s, x_1, s_1)      # residual 1
        x, s = self.norm2(x_2, s_2)
        x, s = self.qact3(x, s)
        x, s = self.mlp(x, s)
        return self.qact4(self.drop_path(x), s, x_2, s_2)          # residual 2


class VisionTransformer(EngineDispatch, nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, patch_embed_bw=8, pos_encoding_bw=8, block_input_bw=8,
                 attention_out_bw=8, softmax_bw=8, mlp_out_bw=8, norm2_in_bw=8, att_block_out_bw=8,
                 gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit"):
        super().__init__()
        if representation_size:
            raise NotImplementedError("pre_logits representation layers are not part of the integer path")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads = depth, num_heads
        self.geometry = (img_size, patch_size, in_chans, float(mlp_ratio), bool(qkv_bias), qk_scale)
        self.op_types = (str(gelu_type).lower(), str(softmax_type).lower(), str(layernorm_type).lower())
        gelu_layer, softmax_cls, norm_layer = get_gelu(gelu_type), get_softmax(softmax_type), get_layernorm(layernorm_type)

        self.qact_input = QuantAct()
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      bitwidth_out=patch_embed_bw)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.qact_pos = QuantAct(pos_encoding_bw)
        self.qact1 = QuantAct(block_input_bw)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], act_layer=gelu_layer,
                  norm_layer=norm_layer, softmax_cls=softmax_cls, attention_out_bw=attention_out_bw,
                  softmax_bw=softmax_bw, mlp_out_bw=mlp_out_bw, norm2_in_bw=norm2_in_bw,
                  att_block_out_bw=att_block_out_bw) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.qact2 = QuantAct()
        self.pre_logits = nn.Identity()
        self.head = QuantLinear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        self._init_dispatch()

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---------------------------------------------------------------- module-by-module path
    def forward_features(self, x):
        B = x.shape[0]
        x, s = self.qact_input(x)
        x, s = self.patch_embed(x, s)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)   # raw float cls row shares the patch scale
        x_pos, s_pos = self.qact_pos(self.pos_embed)
        x, s = self.qact1(x, s, x_pos, s_pos)
        x = self.pos_drop(x)
        for blk in self.blocks:
            x, s = blk(x, s)
        x, s = self.norm(x, s)
        x, s = self.qact2(x[:, 0], s)
        return self.pre_logits(x), s

    # ---------------------------------------------------------------- fused engine path (dispatch.py)
    def engine_unsupported_reason(self):
        """None when the fused int8 engine computes exactly what this module tree would; else why not."""
        if len(set(self.op_types)) != 1 or self.op_types[0] not in ("ivit", "ibert"):
            return f"operator family {self.op_types} (fused engine: all three operators 'ivit', or all three 'ibert')"
        if self.embed_dim // self.num_heads != 64 or self.embed_dim % 64:
            return "head_dim != 64"
        img, patch = self.geometry[0], self.geometry[1]
        if (self.geometry[2:] != (3, 4.0, True, None) or not isinstance(img, int) or not isinstance(patch, int) or img % patch
                or (3 * patch * patch) % 64 or (img // patch) ** 2 + 1 > 207):
            return (f"geometry {self.geometry} (fused engine: square images, 3 channels, img_size % patch_size == 0, "
                    "3 * patch_size^2 % 64 == 0, at most 207 tokens, mlp_ratio 4, qkv bias)")
        if self.num_classes <= 0:
            return "no classification head"
        # every QuantAct of the DeiT / ViT engine is 8 bit, except the 16-bit one inside IBERTIntSoftmax (ibert_modules.py:247) ...
        inner = {f"blocks.{i}.attn.int_softmax.act": 16 for i in range(self.depth)} if self.op_types[0] == "ibert" else {}
        bad = self._width_mismatch(inner)
        self._engine_widths = (8, 8, 8)        # (stream, softmax, position embedding)
        a, m = self.blocks[0].attn.int_softmax, self.blocks[0].mlp.act
        sm_bits = int(getattr(a, "output_bit", 8))
        if bad:
            # ... or the 16-bit residual stream: patch_embed_bw = block_input_bw = attention_out_bw = mlp_out_bw = norm2_in_bw =
            # att_block_out_bw = 16 (vit_quant.py:180-187), softmax_bw and pos_encoding_bw 8 or 16: engine stream_bits = 16
            w16 = {"patch_embed.qact": 16, "qact1": 16, **inner}
            for i in range(self.depth):
                w16.update({f"blocks.{i}.attn.qact3": 16, f"blocks.{i}.mlp.qact2": 16, f"blocks.{i}.qact2": 16, f"blocks.{i}.qact4": 16})
            for pos_bits in (8, 16):
                if self._width_mismatch({**w16, "qact_pos": pos_bits}) is None and sm_bits in (8, 16):
                    bad, self._engine_widths = None, (16, sm_bits, pos_bits)
        if bad:
            return bad
        if (sm_bits != 8 and self._engine_widths[0] == 8) or getattr(m, "output_bit", 8) != 8:
            return "Shiftmax / ShiftGELU output width != 8"
        return None

    def _build_engine(self, device, max_batch):
        from .engine import IntViTEngine
        return IntViTEngine(dict(self.state_dict()), self.ranges(), self.embed_dim, self.depth, self.num_heads,
                            device=device, max_batch=max_batch, family=self.op_types[0],
                            img_size=self.geometry[0], patch_size=self.geometry[1],
                            **dict(zip(("stream_bits", "softmax_bits", "pos_bits"),
                                       self._engine_widths if self.engine_unsupported_reason() is None else (8, 8, 8))))

    def forward(self, x):
        if self.takes_engine(x):
            _, logits_f32, _ = self.engine(x.shape[0])(x.contiguous().float())
            return logits_f32.clone()
        if not self.is_frozen():
            self.invalidate_engine()     # running-stat QuantActs replace their range buffers: any snapshot is stale
        # a frozen I-ViT model run module by module carries int8 between its modules (quantization_utils/lazy.py)
        with lazy.scope(x.is_cuda and not self.training and self.op_types in (("ivit",) * 3, ("ibert",) * 3) and self.is_frozen()):
            x, s = self.forward_features(x)
            x, _ = self.head(x, s)
        return x.to_float(boundary=True) if isinstance(x, lazy.QT) else x


def _factory(embed_dim, depth, num_heads, name):
    def make(pretrained=False, **kwargs):
        if pretrained:
            raise RuntimeError(f"{name}(pretrained=True) downloads weights (vit_quant.py:325-331); there is no network "
                               "here -- load a state_dict instead")
        return VisionTransformer(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                                 qkv_bias=True, **kwargs)
    make.__name__ = name
    return make


deit_tiny_patch16_224 = _factory(192, 12, 3, "deit_tiny_patch16_224")
deit_small_patch16_224 = _factory(384, 12, 6, "deit_small_patch16_224")
deit_base_patch16_224 = _factory(768, 12, 12, "deit_base_patch16_224")
vit_base_patch16_224 = _factory(768, 12, 12, "vit_base_patch16_224")
vit_large_patch16_224 = _factory(1024, 24, 16, "vit_large_patch16_224")
