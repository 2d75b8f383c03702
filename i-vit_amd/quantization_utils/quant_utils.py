"""The functional layer of the reference's quantisers under its own names
(/root/reference/models/quantization_utils/quant_utils.py:12-267), for code that calls them directly instead of through the
modules: ``symmetric_linear_quantization_params``, ``linear_quantize``, ``SymmetricQuantFunction``, ``floor_ste`` / ``round_ste``,
``batch_frexp``, ``fixedpoint_mul`` -- same arguments, same return conventions (integer-valued float32 tensors, the caller
multiplies by the scale).

The integer arithmetic runs in the HIP kernels the modules use: ``fixedpoint_mul`` is ``ivit_requant_i32`` (dyadic multiplier
from ``batch_frexp``'s Decimal rounding, float64 product, round-half-even, clamp), ``SymmetricQuantFunction`` on an activation
with a per-tensor scale is ``ivit_quantize_input_f32_i32``.  Tensors must live on the GPU; nothing here computes on the host
except what the reference itself computes there (``batch_frexp``: numpy ``frexp`` + ROUND_HALF_UP, quant_utils.py:151-175).
The straight-through backward passes are the reference's (the gradient scaled by 1 / scale); training is out of scope
 but a graph that contains these functions still differentiates.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.autograd import Function

from .. import _lib
from ..prepare import f32

__all__ = ["linear_quantize", "symmetric_linear_quantization_params", "SymmetricQuantFunction", "floor_ste", "round_ste",
           "batch_frexp", "fixedpoint_mul", "softmax"]


def _st():
    return _lib.stream_ptr()


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.IvitError(f"{what}: the tensor must be on the GPU (this package has no CPU path)")


def _broadcast_scale(t: torch.Tensor, ndim: int, is_weight: bool):
    """the reshape table of linear_quantize / SymmetricQuantFunction.backward (quant_utils.py:22-46, 102-119)"""
    if is_weight:
        return t.view(-1, 1, 1, 1) if ndim == 4 else t.view(-1, 1) if ndim == 2 else t.view(-1)
    if ndim == 2:
        return t.view(1, -1)
    if ndim == 3:
        return t.view(1, 1, -1)
    if ndim == 4:
        return t.view(1, -1, 1, 1)
    raise NotImplementedError


def linear_quantize(input, scale, zero_point, is_weight):
    """round(1 / scale * input + zero_point) with the reference's broadcasting of `scale` (quant_utils.py:12-49)."""
    scale = _broadcast_scale(scale, input.dim(), is_weight)
    zero_point = _broadcast_scale(zero_point, input.dim(), is_weight) if zero_point.numel() > 1 else zero_point
    return torch.round(1. / scale * input + zero_point)


def symmetric_linear_quantization_params(num_bits, min_val, max_val):
    """scale = max(-min, max) / (2^(bits-1) - 1), clamped at float32 eps (quant_utils.py:52-70)."""
    with torch.no_grad():
        n = 2 ** (num_bits - 1) - 1
        eps = torch.finfo(torch.float32).eps
        max_val = torch.max(-min_val, max_val)
        scale = max_val / float(n)
        scale = scale.clamp(min=eps)
    return scale


class SymmetricQuantFunction(Function):
    """clamp(round(1 / scale * x), -2^(k-1), 2^(k-1) - 1) (quant_utils.py:73-119); integer-valued float32 out."""

    @staticmethod
    def forward(ctx, x, k, specified_scale, is_weight):
        _need_gpu(x, "SymmetricQuantFunction")
        scale = specified_scale
        ctx.scale, ctx.is_weight = scale, is_weight
        n = 2 ** (k - 1) - 1
        if not is_weight and scale.numel() == 1 and 2 <= k <= 32:
            xin = x.detach().contiguous().float()
            q = torch.empty(xin.shape, dtype=torch.int32, device=x.device)
            inv = float(f32(1.0) / f32(scale.reshape(-1)[0].item()))
            _lib.call("ivit_quantize_input_f32_i32", _lib.ptr(xin), _lib.ptr(q), xin.numel(), inv, int(k), _st())
            return q.to(torch.float32)
        # weights (per-output-channel scales): done once per fix(); the modules use prepare.LinearParams for it
        zero_point = torch.zeros((), device=x.device)
        return torch.clamp(linear_quantize(x, scale, zero_point, is_weight=is_weight), -n - 1, n)

    @staticmethod
    def backward(ctx, grad_output):
        scale = _broadcast_scale(ctx.scale, grad_output.dim(), ctx.is_weight)
        return grad_output.clone() / scale, None, None, None


class floor_ste(Function):
    """straight-through estimator of torch.floor (quant_utils.py:122-133)"""

    @staticmethod
    def forward(ctx, x):
        return torch.floor(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.clone()


class round_ste(Function):
    """straight-through estimator of torch.round (quant_utils.py:136-147)"""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.clone()


def batch_frexp(inputs, max_bit=31):
    """(mantissa, exponent) of a tensor of scale ratios (quant_utils.py:151-175): m = ROUND_HALF_UP(frexp mantissa * 2^max_bit)
    as int64, e = max_bit - frexp exponent as float -- on the device of `inputs`, in its shape."""
    shape = inputs.size()
    v = inputs.detach().reshape(-1).cpu().numpy()
    mant, ex = np.frexp(v)
    m = np.floor(mant.astype(np.float64) * 2.0 ** max_bit + 0.5).astype(np.int64)     # positive ratios: half up == floor(. + 0.5)
    neg = mant < 0
    if neg.any():                                                                      # Decimal ROUND_HALF_UP rounds half away from zero
        m[neg] = -np.floor(-mant[neg].astype(np.float64) * 2.0 ** max_bit + 0.5).astype(np.int64)
    e = (float(max_bit) - ex).astype(np.float64 if v.dtype == np.float64 else np.float32)
    return torch.from_numpy(m).to(inputs.device).view(shape), torch.from_numpy(e).to(inputs.device).view(shape)


class fixedpoint_mul(Function):
    """The requantisation every QuantAct performs (quant_utils.py:178-261): z = round(pre_act / s_pre); out =
    RNE(float64(z) * m / 2^e) with (m, e) = batch_frexp(double(s_pre) / double(float(s_z))), plus the same for the identity
    branch, clamped to the `bit_num` range.  Integer-valued float32 out (the caller multiplies by z_scaling_factor)."""

    @staticmethod
    def forward(ctx, pre_act, pre_act_scaling_factor, bit_num, quant_mode, z_scaling_factor, identity=None,
                identity_scaling_factor=None):
        from .quant_modules import _me_tables, to_int32
        _need_gpu(pre_act, "fixedpoint_mul")
        if pre_act.dim() not in (2, 3):
            raise NotImplementedError("fixedpoint_mul: channel-last 2-D / 3-D activations (the layouts the ViT / Swin paths use)")
        if quant_mode != "symmetric":
            raise NotImplementedError("fixedpoint_mul: only quant_mode='symmetric' (the only mode the reference's models use)")
        ctx.identity, ctx.z_scaling_factor = identity, z_scaling_factor
        s_z = f32(z_scaling_factor.reshape(-1)[0].item())
        C = pre_act.shape[-1]
        z = to_int32(pre_act, pre_act_scaling_factor)
        m, e, n_me = _me_tables(pre_act_scaling_factor, s_z, pre_act.device)
        z2 = m2 = e2 = None
        n2 = 0
        if identity is not None:
            z2 = to_int32(identity.expand_as(pre_act) if identity.shape != pre_act.shape else identity, identity_scaling_factor)
            m2, e2, n2 = _me_tables(identity_scaling_factor, s_z, pre_act.device)
        bits = bit_num if bit_num in (4, 8, 16, 32) else 32      # other widths: the reference does not clamp (:249-255)
        q = torch.empty(z.shape, dtype=torch.int32, device=pre_act.device)
        _lib.call("ivit_requant_i32", _lib.ptr(z), z.numel() // C, C, _lib.ptr(m), _lib.ptr(e), n_me, _lib.ptr(z2), _lib.ptr(m2),
                  _lib.ptr(e2), n2, int(bits), _lib.ptr(q), _st())
        return q.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_output):
        identity_grad = grad_output.clone() / ctx.z_scaling_factor if ctx.identity is not None else None
        return grad_output.clone() / ctx.z_scaling_factor, None, None, None, None, identity_grad, None


def softmax(x, dim: int, onnx_trace: bool = False):
    """float helper the reference keeps next to the quantisers (quant_utils.py:263-267); not on the integer path"""
    import torch.nn.functional as F
    return F.softmax(x.float(), dim=dim) if onnx_trace else F.softmax(x, dim=dim, dtype=torch.float32)
