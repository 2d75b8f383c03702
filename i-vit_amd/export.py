"""Integer-parameter export of a frozen DeiT/ViT model, and the reverse: an engine parameter source that needs
no float weights.

On-disk format = what /root/reference/TVM_benchmark/convert_model.py:12-148 produces for its TVM runtime:
  * `params.npy`  - a pickled dict (np.save of a dict, as the reference writes it, :66) of int8 weights, int32 biases,
    int32 LayerNorm biases and the float cls / pos tensors under the reference's renamed keys
    (`embed_conv_weight`, `block_%d_attn_qkv_weight`, ..., `head_bias`, `block_%d_norm1_bias`, `norm_bias`,
    `cls_token_weight`, `pos_embed_weight`);
  * the scale table the reference keeps in `QuantizeContext.qconfig_dict` (load_qconfig, :69-148).  The reference never
    writes it to disk (it is rebuilt from the checkpoint at conversion time); here it is stored next to the parameters
    as `qconfig.npy`: for every entry the `input_scale` / `kernel_scale` / `output_scale` fields of the reference's
    QConfig, plus the raw `*scaling_factor` vectors they are derived from, so that the HIP engine can be built from the
    two files alone (`ExportSource`), with no PyTorch model and no float weights.

Everything here is host-side numpy in the reference's float32 arithmetic (prepare.py); nothing touches the GPU.
"""
from __future__ import annotations

import os

import numpy as np

from .prepare import LayerNormParams, LinearParams, f32, sym_scale
from .synth import qact_names

_LINEARS = ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")
# which QuantAct feeds each QuantLinear / QuantConv2d (vit_quant.py:61-90,142-155; layers_quant.py:145-154,197-203)
_INPUT_OF = {"attn.qkv": "qact1", "attn.proj": "attn.qact2", "mlp.fc1": "qact3", "mlp.fc2": "mlp.qact1"}


class FloatSource:
    """Engine parameter source: float state_dict + QuantAct ranges (what a checkpoint holds)."""

    def __init__(self, float_state, ranges):
        self.P = {k: np.asarray(v, dtype=np.float32) for k, v in float_state.items() if np.asarray(v).dtype.kind == "f"}
        self.R = ranges

    def act_scale(self, name, bits=8):
        lo, hi = self.R[name]
        return sym_scale(lo, hi, bits)

    def linear(self, name, s_in):
        return LinearParams(self.P[name + ".weight"], self.P.get(name + ".bias"), s_in)

    def layernorm(self, prefix, s_out):
        return LayerNormParams(self.P[prefix + ".weight"], self.P[prefix + ".bias"], s_out)

    def tensor(self, name):
        return self.P[name]


def export_integer_params(float_state, ranges, depth: int):
    """-> (params, qconfig): the two dicts described in the module docstring, for a DeiT/ViT of `depth` blocks."""
    src = FloatSource(float_state, ranges)
    act = {n: src.act_scale(n) for n in qact_names(depth)}
    params, scales = {}, {}

    def put_linear(ref_name, new_prefix, s_in, conv=False):
        lp = src.linear(ref_name, s_in)
        W = src.tensor(ref_name + ".weight")
        params[new_prefix + "weight"] = lp.W8.reshape(W.shape)
        params[new_prefix + "bias"] = lp.b32.reshape(1, -1, 1, 1) if conv else lp.b32
        scales[ref_name + (".conv_scaling_factor" if conv else ".fc_scaling_factor")] = lp.sw
        return lp

    put_linear("patch_embed.proj", "embed_conv_", act["qact_input"], conv=True)
    for i in range(depth):
        p = f"blocks.{i}."
        for lin in _LINEARS:
            put_linear(p + lin, f"block_{i}_{lin.replace('.', '_')}_", act[p + _INPUT_OF[lin]])
        for nrm, qa in (("norm1", "qact1"), ("norm2", "qact3")):
            ln = src.layernorm(p + nrm, act[p + qa])
            params[f"block_{i}_{nrm}_bias"] = ln.bias_int.astype(np.int32)
            scales[p + nrm + ".norm_scaling_factor"] = ln.s_ln
    put_linear("head", "head_", act["qact2"])
    ln = src.layernorm("norm", act["qact2"])
    params["norm_bias"] = ln.bias_int.astype(np.int32)
    scales["norm.norm_scaling_factor"] = ln.s_ln
    params["cls_token_weight"] = src.tensor("cls_token")
    params["pos_embed_weight"] = src.tensor("pos_embed")
    for n, s in act.items():
        scales[n + ".act_scaling_factor"] = f32(s)
    # derived module outputs the reference also records (quant_modules.py:409; ivit_modules.py:124,176)
    for i in range(depth):
        p = f"blocks.{i}."
        s_a1 = act[p + "attn.qact1"]
        scales[p + "attn.matmul_1.act_scaling_factor"] = f32(s_a1 * s_a1)
        scales[p + "attn.int_softmax.act_scaling_factor"] = f32(1.0 / 128.0)
        scales[p + "attn.matmul_2.act_scaling_factor"] = f32(f32(1.0 / 128.0) * s_a1)
        scales[p + "mlp.act.act_scaling_factor"] = f32(act[p + "mlp.qact_gelu"] * f32(1.0 / 128.0))
    return params, build_qconfig(scales, depth)


def build_qconfig(scales: dict, depth: int) -> dict:
    """load_qconfig of convert_model.py:69-148 as a plain dict: entry -> {input_scale, kernel_scale, output_scale,
    input_dtype}; the raw scale vectors are kept under 'scales'."""
    S = scales
    qc = {}

    def entry(input_scale=8.0, kernel_scale=8.0, output_scale=74.0, input_dtype="int8"):
        # the twelve fields of the reference's QConfig namedtuple with its defaults (TVM_benchmark/models/layers.py:10-14)
        return dict(from_dtype="int32", from_scale=65.0, from_zero_point=0.0, input_dtype=input_dtype,
                    input_scale=input_scale, input_zero_point=0.0, kernel_dtype="int8", kernel_scale=kernel_scale,
                    kernel_zero_point=0.0, output_dtype="int32", output_scale=output_scale, output_zero_point=0.0)

    def lin(in_act, kernel):
        i, k = S[in_act + ".act_scaling_factor"], S[kernel]
        return entry(i, k, (np.asarray(i, np.float32) * k).astype(np.float32))

    A = lambda n: S[n + ".act_scaling_factor"]  # noqa: E731
    qc["qconfig_pos"] = entry(output_scale=A("qact_pos"))
    qc["qconfig_addpos"] = entry(A("patch_embed.qact"), 8.0, A("qact1"), "int16")
    qc["qconfig_embed_conv"] = lin("qact_input", "patch_embed.proj.conv_scaling_factor")
    last = A("qact1")
    for i in range(depth):
        p = f"blocks.{i}."
        qc[f"block_{i}_qconfig_norm1"] = entry(last, 8.0, S[p + "norm1.norm_scaling_factor"])
        qc[f"block_{i}_qconfig_qkv"] = lin(p + "qact1", p + "attn.qkv.fc_scaling_factor")
        qc[f"block_{i}_qconfig_matmul_1"] = entry(A(p + "attn.qact1"), 8.0, A(p + "attn.matmul_1"))
        qc[f"block_{i}_qconfig_softmax"] = entry(A(p + "attn.qact_attn1"), 8.0, A(p + "attn.int_softmax"))
        qc[f"block_{i}_qconfig_matmul_2"] = entry(A(p + "attn.int_softmax"), 8.0, A(p + "attn.matmul_2"))
        qc[f"block_{i}_qconfig_proj"] = lin(p + "attn.qact2", p + "attn.proj.fc_scaling_factor")
        qc[f"block_{i}_qconfig_add1"] = entry(A(p + "attn.qact3"), 8.0, A(p + "qact2"), "int16")
        qc[f"block_{i}_qconfig_norm2"] = entry(A(p + "qact2"), 8.0, S[p + "norm2.norm_scaling_factor"])
        qc[f"block_{i}_qconfig_fc1"] = lin(p + "qact3", p + "mlp.fc1.fc_scaling_factor")
        qc[f"block_{i}_qconfig_gelu"] = entry(A(p + "mlp.qact_gelu"), 8.0, A(p + "mlp.act"), "int8")
        qc[f"block_{i}_qconfig_fc2"] = lin(p + "mlp.qact1", p + "mlp.fc2.fc_scaling_factor")
        qc[f"block_{i}_qconfig_add2"] = entry(A(p + "mlp.qact2"), 8.0, A(p + "qact4"), "int16")
        last = A(p + "qact4")
    qc["qconfig_norm"] = entry(A(f"blocks.{depth - 1}.mlp.qact2"), 8.0, S["norm.norm_scaling_factor"])   # :140-141 reuses the loop's last input_scale
    qc["qconfig_head"] = lin("qact2", "head.fc_scaling_factor")
    qc["scales"] = dict(S)
    qc["depth"] = depth
    return qc


def save_export(params: dict, qconfig: dict, save_path: str):
    os.makedirs(save_path, exist_ok=True)
    np.save(os.path.join(save_path, "params.npy"), params)      # pickled dict, like convert_model.py:66
    np.save(os.path.join(save_path, "qconfig.npy"), qconfig)


def load_export(save_path: str):
    params = np.load(os.path.join(save_path, "params.npy"), allow_pickle=True).item()
    qconfig = np.load(os.path.join(save_path, "qconfig.npy"), allow_pickle=True).item()
    return params, qconfig


def save_params_from_state_dict(state_dict, depth: int, save_path: str):
    """convert_model.py:12-66 applied to a state_dict of the module mirror after a frozen forward (the
    `weight_integer` / `bias_integer` buffers are then populated, as in the reference)."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dict.items()}
    out = {"embed_conv_weight": sd["patch_embed.proj.weight_integer"].astype(np.int8),
           "embed_conv_bias": sd["patch_embed.proj.bias_integer"].astype(np.int32).reshape(1, -1, 1, 1)}
    for i in range(depth):
        for lin in _LINEARS:
            out[f"block_{i}_{lin.replace('.', '_')}_weight"] = sd[f"blocks.{i}.{lin}.weight_integer"].astype(np.int8)
            out[f"block_{i}_{lin.replace('.', '_')}_bias"] = sd[f"blocks.{i}.{lin}.bias_integer"].astype(np.int32)
        for nrm in ("norm1", "norm2"):
            out[f"block_{i}_{nrm}_bias"] = sd[f"blocks.{i}.{nrm}.bias_integer"].astype(np.int32)
    out["head_weight"] = sd["head.weight_integer"].astype(np.int8)
    out["head_bias"] = sd["head.bias_integer"].astype(np.int32)
    out["norm_bias"] = sd["norm.bias_integer"].astype(np.int32)
    out["cls_token_weight"] = sd["cls_token"]
    out["pos_embed_weight"] = sd["pos_embed"]
    os.makedirs(save_path, exist_ok=True)
    np.save(os.path.join(save_path, "params.npy"), out)
    return out


class _Lin:
    """LinearParams look-alike built from exported integers."""

    def __init__(self, W8, b32, sw, s_in):
        self.W8 = np.ascontiguousarray(W8.reshape(W8.shape[0], -1).astype(np.int8))
        self.b32 = None if b32 is None else np.ascontiguousarray(b32.reshape(-1).astype(np.int32))
        self.sw = np.asarray(sw, np.float32)
        self.s_acc = (self.sw * f32(s_in)).astype(np.float32)
        self.K = self.W8.shape[1]

    requant_to = LinearParams.requant_to


class _Ln:
    def __init__(self, bias_int, s_ln, s_out):
        from .prepare import dyadic
        self.bias_int = np.asarray(bias_int, np.float32)
        self.s_ln = np.asarray(s_ln, np.float32)
        self.m, self.e = dyadic(self.s_ln, s_out)
        if np.any(self.e < 40):
            raise ValueError("LayerNorm requantiser with multiplier > 2^-9 (e < 40) is outside the kernel's contract")


class ExportSource:
    """Engine parameter source: the exported integer parameters + scale table (no float weights)."""

    _RENAME = {"patch_embed.proj": "embed_conv_", "head": "head_", "norm": "norm_"}

    def __init__(self, params: dict, qconfig: dict):
        self.P, self.S = params, qconfig["scales"]
        self.depth = qconfig["depth"]

    def _prefix(self, name):
        if name in self._RENAME:
            return self._RENAME[name]
        blk, rest = name.split(".", 2)[1:]
        return f"block_{blk}_{rest.replace('.', '_')}_"

    def act_scale(self, name, bits=8):
        return f32(self.S[name + ".act_scaling_factor"])

    def linear(self, name, s_in):
        pre = self._prefix(name)
        key = name + (".conv_scaling_factor" if name == "patch_embed.proj" else ".fc_scaling_factor")
        return _Lin(self.P[pre + "weight"], self.P[pre + "bias"], self.S[key], s_in)

    def layernorm(self, prefix, s_out):
        return _Ln(self.P[self._prefix(prefix) + "bias"], self.S[prefix + ".norm_scaling_factor"], s_out)

    def tensor(self, name):
        return np.asarray(self.P[{"cls_token": "cls_token_weight", "pos_embed": "pos_embed_weight"}[name]], np.float32)
