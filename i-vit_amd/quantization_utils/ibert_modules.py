"""I-BERT operator family behind the reference's module API
(/root/reference/models/quantization_utils/ibert_modules.py:12-319; registry key 'ibert', the fork's default,
vit_quant.py:188-190), backed by the HIP kernels of csrc/ibert.hip.

Same class names, constructor arguments, parameters / buffers (state_dict keys) and `forward(x, scaling_factor) ->
(y, scaling_factor)` contract.  The scalar constants of each call (floor(coef / scale), output scales) are evaluated
on the host in float32 with the reference's own sequence of operations; the tensor work runs in the kernels.
Only the integer ('symmetric') mode exists on this path: `quant_mode='none'` / `force_dequant` (float GELU / softmax /
LayerNorm) raise.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..prepare import EPS32, dyadic, f32
from . import lazy
from .quant_modules import QuantAct, _dev_table, _st, to_float, to_int32


# ---- host-side scalar constants of the operators, float32 with the reference's own sequence of operations; shared by the
#      modules below and by the fused engine (engine.py, family "ibert")
_GELU_K, _GELU_N = 1.4142, 6
_GELU_COEFF = (-0.2888, -1.769, 1 / -0.2888)          # :172-175 (coeff[2] /= coeff[0])
_SM_X0, _SM_N = -0.6931, 30
_SM_COEF = (0.35815147, 0.96963238 / 0.35815147, 1.0 / 0.35815147)   # :253-257


def gelu_constants(s):
    """(b_int, c_int, shift_int, s_out) in float32, ibert_modules.py:205-206, 214-216, 229, 232."""
    s = f32(s)
    sf = f32(s / f32(_GELU_K))
    b_int = np.floor(f32(f32(_GELU_COEFF[1]) / sf))
    c_int = np.floor(f32(f32(_GELU_COEFF[2]) / f32(sf * sf)))
    sf2 = f32(f32(f32(sf * sf) * f32(_GELU_COEFF[0])) * f32(2 ** _GELU_N))
    shift_int = np.floor(f32(f32(1.0) / sf2))
    return float(b_int), float(c_int), float(shift_int), f32(f32(s * sf2) / f32(2))


def softmax_constants(s, lo, hi):
    """(x0_int, b_int, c_int, exp_sf, act_sf, m, e): :277-294 and the internal 16-bit QuantAct (range lo..hi) of :308."""
    s = f32(s)
    x0_int = np.floor(f32(f32(_SM_X0) / s))                                    # :287
    b_int = np.floor(f32(f32(_SM_COEF[1]) / s))                                # :277
    c_int = np.floor(f32(f32(_SM_COEF[2]) / f32(s * s)))                       # :278
    exp_sf = f32(f32(f32(_SM_COEF[0]) * f32(s * s)) / f32(2 ** _SM_N))         # :282, 294
    act_sf = max(f32(f32(max(-f32(lo), f32(hi))) / f32(2 ** 15 - 1)), f32(EPS32))   # quant_utils.py:52-70, 16 bit
    m, e = dyadic(exp_sf, act_sf)
    return float(x0_int), float(b_int), float(c_int), exp_sf, act_sf, int(m[0]), int(e[0])


def _check_mode(quant_mode, force_dequant, what):
    if quant_mode != "symmetric" or force_dequant in ("nonlinear", what):
        raise NotImplementedError(f"IBERT {what}: only the integer (quant_mode='symmetric') form runs on the MI355X path")


class IBERTIntLayerNorm(nn.Module):
    """ibert_modules.py:12-158."""

    def __init__(self, normalized_shape, output_bit=8, overflow_handling=True, quant_mode="symmetric",
                 force_dequant="none", elementwise_affine=True, eps=1e-5, use_int_sqrt=False):
        super().__init__()
        _check_mode(quant_mode, force_dequant, "layernorm")
        if use_int_sqrt:
            raise NotImplementedError("IBERT layernorm: use_int_sqrt=True (Newton integer sqrt) is not implemented")
        self.quant_mode, self.overflow_handling, self.use_int_sqrt = quant_mode, overflow_handling, use_int_sqrt
        self.register_buffer("shift", torch.zeros(1))
        self.output_bit, self.dim_sqrt, self.eps = output_bit, None, eps
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = normalized_shape
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self._cache = None

    def fix(self):
        self.overflow_handling = False

    def unfix(self):
        self.overflow_handling = True

    def forward(self, x, scaling_factor=None, exponents=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(scaling_factor)
            if x.q8 is not None and s_in is not None and s_in.size == 1 and not self.overflow_handling:
                def build():      # the scale this module returns: sqrt(C) / 2^30 * gamma (:145-153)
                    C_ = x.shape[-1]
                    sf = f32(np.sqrt(f32(C_)).astype(np.float32) / f32(2 ** 30))
                    return lazy.QS.make((sf * self.weight.detach().cpu().numpy().astype(np.float32)).astype(np.float32), x.device)
                s_ln = lazy._cache(self, ("s_ln", self.weight._version, str(x.device)), build)
                return lazy.pending("ibln", self, x.shape, x.device, (x,), (scaling_factor,), s_ln)
            x = x.to_float()
        return self._slow(x, scaling_factor, exponents)

    def _slow(self, x, scaling_factor=None, exponents=None):
        C = x.shape[-1]
        key = (self.weight._version, self.bias._version, x.device)
        if self._cache is None or self._cache[0] != key:
            gamma = self.weight.detach().cpu().numpy().astype(np.float32)
            beta = self.bias.detach().cpu().numpy().astype(np.float32)
            sf = f32(np.sqrt(f32(C)).astype(np.float32) / f32(2 ** 30))            # :124, 145
            bias_int = np.floor(((beta / gamma).astype(np.float32) / sf).astype(np.float32))   # :148-149
            self._cache = (key, _dev_table(bias_int.astype(np.float32), x.device),
                           _dev_table((sf * gamma).astype(np.float32), x.device))
            self.dim_sqrt = torch.sqrt(torch.tensor(float(C)))
        _, bias_int, s_out = self._cache
        xin = x.contiguous().float()
        s_in = scaling_factor.reshape(-1).contiguous().float()
        assert s_in.numel() in (1, C)
        if self.overflow_handling:
            k = to_int32(x, scaling_factor)
            # training / calibration-time guard of :134-137: raise the shift until sum(y^2) < 2^32
            y = k.double() - torch.round(k.double().mean(dim=-1, keepdim=True))
            var = (torch.floor(y / 2.0 ** float(self.shift)) ** 2).sum(dim=-1)
            if float(var.max()) >= 2.0 ** 32:
                var0 = (y ** 2).sum(dim=-1)
                shift = torch.log2(torch.sqrt(var0 / 2 ** 32)).ceil().max()
                self.shift = torch.max(self.shift, shift.to(self.shift).reshape(1))
        # the literal kernel: x / scaling_factor, float32 mean and variance sums in torch's reduction order, ... (:126-153) for
        # any input scale (csrc/ibert.hip, second half)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _lib.call("ivit_ibert_layernorm_f32_f32", _lib.ptr(xin), C, xin.numel() // C, C, _lib.ptr(s_in), s_in.numel(),
                  _lib.ptr(bias_int), _lib.ptr(s_out), float(2.0 ** float(self.shift)), _lib.ptr(out), C, _st())
        return out, s_out


class IBERTIntGELU(nn.Module):
    """ibert_modules.py:161-235."""

    def __init__(self, quant_mode="symmetric", force_dequant="none"):
        super().__init__()
        _check_mode(quant_mode, force_dequant, "gelu")
        self.register_buffer("input_scaling_factor", torch.ones(1))
        self.quant_mode = quant_mode
        self.k, self.n = 1.4142, 6
        self.coeff = [-0.2888, -1.769, 1]
        self.coeff[2] /= self.coeff[0]

    def fix(self):
        pass

    def unfix(self):
        pass

    def constants(self, s):
        """(b_int, c_int, shift_int, s_out) in float32, :205-206, 214-216, 229, 232."""
        return gelu_constants(s)

    def forward(self, x, scaling_factor=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(scaling_factor)
            if x.q8 is not None and s_in is not None and s_in.size == 1:
                so = lazy._cache(self, ("s_out", s_in.tobytes(), str(x.device)), lambda: lazy.QS.make(f32(self.constants(s_in[0])[3]), x.device))
                return lazy.pending("ibgelu", self, x.shape, x.device, (x,), (scaling_factor,), so)
            x = x.to_float()
        return self._slow(x, scaling_factor)

    def _slow(self, x, scaling_factor=None):
        s = float(scaling_factor.reshape(-1)[0])
        b_int, c_int, shift_int, s_out = self.constants(s)
        xin = x.contiguous().float()
        out = torch.empty(xin.shape, dtype=torch.float32, device=x.device)
        _lib.call("ivit_ibert_gelu_f32_f32", _lib.ptr(xin), xin.numel(), s, b_int, c_int, shift_int, float(s_out),
                  _lib.ptr(out), _st())                                            # literal: :226-234 on x / s itself
        so = torch.full((1,), float(s_out), dtype=torch.float32, device=x.device)
        return out, so


class IBERTIntSoftmax(nn.Module):
    """ibert_modules.py:237-319."""

    def __init__(self, output_bit, quant_mode="symmetric", force_dequant="none"):
        super().__init__()
        _check_mode(quant_mode, force_dequant, "softmax")
        self.output_bit, self.quant_mode = output_bit, quant_mode
        self.act = QuantAct(16, quant_mode=quant_mode)
        self.x0, self.n = -0.6931, 30
        self.coef = [0.35815147, 0.96963238, 1.0]
        self.coef[1] /= self.coef[0]
        self.coef[2] /= self.coef[0]

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor):
        if isinstance(x, lazy.QT):
            if (isinstance(x.node, lazy.Scores) and not x.views and self.output_bit == 8 and not self.act.running_stat
                    and scaling_factor is x.node.s_out_qs):
                so = lazy._cache(self, ("s_out", str(x.device)), lambda: lazy.QS.make(f32(2 / 2 ** self.output_bit), x.device))   # :317
                return lazy.QT.wrap(x.shape, x.device, node=lazy.Probs(x, self)), so
            x = x.to_float()
        return self._slow(x, scaling_factor)

    def _slow(self, x, scaling_factor):
        s = f32(float(scaling_factor.reshape(-1)[0]))
        L = x.shape[-1]
        x0_int, b_int, c_int, exp_sf = softmax_constants(s, -1.0, 1.0)[:4]         # :277-294 (the range-dependent ones below)
        xin = x.contiguous().float()                                               # literal kernel: x / s itself (:303)
        rows = xin.numel() // L
        st = _st()
        if self.act.running_stat:
            # the internal QuantAct(16) observes exp_int (:308): one extra pass that only produces exp_int
            ex = torch.empty(xin.shape, dtype=torch.float32, device=x.device)
            _lib.call("ivit_ibert_softmax_f32_f32", _lib.ptr(xin), L, rows, L, float(s), float(x0_int), float(b_int), float(c_int),
                      float(exp_sf), 1.0, 1 << 30, 30, self.output_bit, None, L, _lib.ptr(ex), st)
            self.act._observe(ex)
        lo, hi = float(self.act.x_min.reshape(-1)[0]), float(self.act.x_max.reshape(-1)[0])
        act_sf, m, e = softmax_constants(s, lo, hi)[4:]                                 # quant_utils.py:52-70, 16 bit
        self.act.act_scaling_factor = torch.full((1,), float(act_sf), dtype=torch.float32, device=x.device)
        out = torch.empty(xin.shape, dtype=torch.float32, device=x.device)
        _lib.call("ivit_ibert_softmax_f32_f32", _lib.ptr(xin), L, rows, L, float(s), float(x0_int), float(b_int), float(c_int),
                  float(exp_sf), float(act_sf), m, e, self.output_bit, _lib.ptr(out), L, None, st)
        so = torch.tensor([2 / 2 ** self.output_bit], dtype=torch.float32, device=x.device)   # :317
        return out, so
