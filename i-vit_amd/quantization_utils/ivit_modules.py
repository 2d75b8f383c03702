"""I-ViT non-linear operators behind the reference's module API
(/root/reference/models/quantization_utils/ivit_modules.py:10-179), backed by the HIP kernels."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..prepare import LayerNormParams, f32
from . import lazy
from .quant_modules import _dev_table, _st, narrow_i8, to_float, to_int32


class IVITIntLayerNorm(nn.LayerNorm):
    """I-LayerNorm, ivit_modules.py:10-65."""

    def __init__(self, normalized_shape, eps=1e-6, elementwise_affine=True):
        super().__init__(normalized_shape, eps, elementwise_affine)
        self.dim_sqrt = None
        self.register_buffer("norm_scaling_factor", torch.zeros(normalized_shape))
        self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._cache = None

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(scaling_factor)
            if x.q8 is not None and s_in is not None and s_in.size == 1:
                def build():
                    lp = LayerNormParams(self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(), f32(1.0))
                    self.bias_integer = _dev_table(lp.bias_int, x.device)       # :59
                    s = lazy.QS.make(lp.s_ln, x.device)
                    self.norm_scaling_factor = s.as_subclass(torch.Tensor)      # :64
                    return s
                s_ln = lazy._cache(self, ("s_ln", self.weight._version, self.bias._version, str(x.device)), build)
                return lazy.pending("ln", self, x.shape, x.device, (x,), (scaling_factor,), s_ln)
            x = x.to_float()
        return self._slow(x, scaling_factor)

    def _slow(self, x, scaling_factor=None):
        C = x.shape[-1]
        key = (self.weight._version, self.bias._version, x.device)
        if self._cache is None or self._cache[0] != key:
            lp = LayerNormParams(self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(), f32(1.0))
            self._cache = (key, _dev_table(lp.bias_int, x.device), _dev_table(lp.s_ln, x.device))
            self.dim_sqrt = torch.sqrt(torch.tensor(float(C)))
        _, bias_int, s_ln = self._cache
        # the literal kernel: x / scaling_factor, the float32 mean in torch's reduction order, .to(int32), ... (:36-63) for
        # any input scale (a calibrated, non power-of-two scale makes x / s a non-integer float, see csrc/literal.hip)
        # x_int.mean(axis=2) (:37) over a TRANSPOSED view (the Swin patch embedding, layers_quant.py:198: the reduced dimension is
        # not the contiguous one) runs through ATen's outer-reduction cascade; rows whose mean is an exact .5 tie depend on it
        outer = 0
        if x.dim() >= 2 and x.stride(-1) != 1 and x.shape[-1] > 1:
            if x.dim() == 3 and x.stride(1) == 1 and x.stride(2) == x.shape[1] and x.stride(0) == x.shape[1] * x.shape[2]:
                outer = x.shape[1]       # contiguous extent of the view
            elif not getattr(IVITIntLayerNorm, "_warned_strided", False):
                # any other layout (4-D, channels-last, a sliced view): the contiguous-order kernel.  Only rows whose float32 mean
                # is an exact .5 tie can depend on the order of the reference's reduction over such a view, and that order is not
                # restated (nor fixed: above 32768 outputs ATen partitions it by thread).
                IVITIntLayerNorm._warned_strided = True
                import warnings
                warnings.warn("IVITIntLayerNorm over a last dimension with stride != 1 that is not the [B, L, C] transpose of a "
                              "contiguous [B, C, L] tensor: evaluated in contiguous order; rows whose mean is an exact .5 tie may "
                              "differ from the reference's strided reduction", RuntimeWarning, stacklevel=3)
        xin = x.contiguous().float()
        s_in = scaling_factor.reshape(-1).contiguous().float()
        assert s_in.numel() in (1, C)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _lib.call("ivit_layernorm_f32_f32_ex", _lib.ptr(xin), C, xin.numel() // C, C, _lib.ptr(s_in), s_in.numel(),
                  _lib.ptr(bias_int), _lib.ptr(s_ln), _lib.ptr(out), C, outer, _st())
        self.bias_integer = bias_int                            # :59
        self.norm_scaling_factor = s_ln                         # :64
        if outer:     # the reference's output is a chain of elementwise ops on x_int: it keeps the input's (transposed) strides
            out = torch.empty_like(x, dtype=torch.float32).copy_(out)
        return out, s_ln


class IVITIntGELU(nn.Module):
    """ShiftGELU, ivit_modules.py:68-126."""

    def __init__(self, output_bit=8, n=23):
        super().__init__()
        if output_bit != 8 or n != 23:
            raise NotImplementedError("the HIP ShiftGELU implements output_bit=8, n=23 (the reference defaults)")
        self.output_bit, self.n = output_bit, n
        self.register_buffer("act_scaling_factor", torch.zeros(1))

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(scaling_factor)
            if x.q8 is not None and s_in is not None and s_in.size == 1:
                s = lazy._cache(self, ("s_out", s_in.tobytes(), str(x.device)),
                                lambda: lazy.QS.make(f32(s_in[0] * f32(1 / 2 ** (self.output_bit - 1))), x.device))   # :121,124
                self.act_scaling_factor = s.as_subclass(torch.Tensor)
                return lazy.pending("gelu", self, x.shape, x.device, (x,), (scaling_factor,), s)
            x = x.to_float()
        return self._slow(x, scaling_factor)

    def _slow(self, x, scaling_factor=None):
        L = x.shape[-1]
        k8 = narrow_i8(to_int32(x, scaling_factor, trunc=True), "IVITIntGELU input")   # :106-107
        out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
        _lib.call("ivit_shiftgelu_i8_i32", _lib.ptr(k8), L, k8.numel() // L, L, float(scaling_factor.reshape(-1)[0]),
                  _lib.ptr(out), L, _st())
        s = (scaling_factor.reshape(-1)[:1].float() * torch.tensor([1 / 2 ** (self.output_bit - 1)],
                                                                  device=x.device)).float()  # :121,124
        self.act_scaling_factor = s
        return to_float(out, s), s


class IVITIntSoftmax(nn.Module):
    """Shiftmax, ivit_modules.py:129-179."""

    def __init__(self, output_bit=8):
        super().__init__()
        if not 2 <= output_bit <= 16:
            raise NotImplementedError("the HIP Shiftmax implements output_bit 2..16")
        self.output_bit, self.n = output_bit, 15
        self.register_buffer("act_scaling_factor", torch.zeros(1))

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor):
        if isinstance(x, lazy.QT):
            if isinstance(x.node, lazy.Scores) and not x.views and self.output_bit == 8 and scaling_factor is x.node.s_out_qs:
                s = lazy._cache(self, ("s_out", str(x.device)), lambda: lazy.QS.make(f32(1 / 2 ** (self.output_bit - 1)), x.device))  # :176
                self.act_scaling_factor = s.as_subclass(torch.Tensor)
                return lazy.QT.wrap(x.shape, x.device, node=lazy.Probs(x, self)), s
            x = x.to_float()
        return self._slow(x, scaling_factor)

    def _slow(self, x, scaling_factor):
        L = x.shape[-1]
        # the literal kernel: the reference discards its .to(int32) (:166) and runs the float32 sequence on x / s itself
        # (Swin's masked scores, swin_quant.py:151-156, included), csrc/literal.hip
        xin = x.contiguous().float()
        if self.output_bit == 8:
            out8 = torch.empty(x.shape, dtype=torch.int8, device=x.device)
            _lib.call("ivit_shiftmax_f32_i8", _lib.ptr(xin), L, xin.numel() // L, L, float(scaling_factor.reshape(-1)[0]),
                      _lib.ptr(out8), L, _st())
        else:     # the softmax_bw knob (vit_quant.py:184): the same sequence with floor(. / 2^(31 - output_bit + 1)) (:175)
            out8 = torch.empty(x.shape, dtype=torch.int16, device=x.device)
            _lib.call("ivit_shiftmax_f32_i16", _lib.ptr(xin), L, xin.numel() // L, L, float(scaling_factor.reshape(-1)[0]),
                      self.output_bit, _lib.ptr(out8), L, _st())
        s = torch.tensor([1 / 2 ** (self.output_bit - 1)], dtype=torch.float32, device=x.device)  # :176
        self.act_scaling_factor = s
        return to_float(out8.to(torch.int32), s), s
