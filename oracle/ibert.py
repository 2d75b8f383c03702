"""CPU restatement (numpy float32, op for op) of the reference's I-BERT operator family
(/root/reference/models/quantization_utils/ibert_modules.py).  TEST INFRASTRUCTURE ONLY, like the rest of oracle/.

The reference evaluates these operators in float32 tensors; every step below is the same IEEE float32 operation on
the same operands in the same order, so the results are bit-identical wherever the reference itself is deterministic:
all steps are element-wise except three row sums, which are sums of integer-valued float32 numbers and therefore
exact (order-free) while they stay below 2^24 -- `ln_inexact_rows` / `softmax_inexact_rows` count the rows where
that does not hold.

Pinned by tests/golden/ibert_kat.npz (outputs of the reference's own modules) and tests/golden/deit_tiny_ibert.npz.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
EPS32 = np.finfo(np.float32).eps


def _dyadic(pre_sf, z_sf):
    """batch_frexp of quant_utils.py:151-175 for one ratio (float64 frexp, round-half-up mantissa)."""
    ns = np.float64(f32(pre_sf)) / np.float64(f32(z_sf))
    mant, ex = np.frexp(ns)
    return np.floor(mant * 2.0 ** 31 + 0.5), 31 - int(ex)


def gelu_constants(s):
    """Scalar constants of IBERTIntGELU.forward for input scale s (ibert_modules.py:172-235): (b_int, c_int, shift_int,
    s_out), all float32; python floats meet float32 tensors exactly as in the reference (the scalar is rounded to float32
    first)."""
    s = f32(s)
    sf = f32(s / f32(1.4142))                                   # :227 scaling_factor / self.k
    c0, c1 = -0.2888, -1.769
    c2 = 1 / c0                                                 # :193 self.coeff[2] /= self.coeff[0] (python floats)
    b_int = np.floor(f32(f32(c1) / sf))                         # :205
    c_int = np.floor(f32(f32(c2) / f32(sf * sf)))               # :206
    sf2 = f32(f32(f32(sf * sf) * f32(c0)) * f32(2 ** 6))        # :214-216
    shift_int = np.floor(f32(f32(1.0) / sf2))                   # :229
    s_out = f32(f32(s * sf2) / f32(2))                          # :232
    return f32(b_int), f32(c_int), f32(shift_int), s_out


def gelu(k, s):
    """k: integers (any int dtype / integer-valued float32) with scale s -> (out_int float32, s_out).  :220-235"""
    s = f32(s)
    b_int, c_int, shift_int, s_out = gelu_constants(s)
    x_int = ((np.asarray(k).astype(f32) * s).astype(f32) / s).astype(f32)      # the float view and :226
    sign = np.sign(x_int).astype(f32)                                           # :209
    abs_int = np.minimum(np.abs(x_int), -b_int).astype(f32)                     # :210-211
    y = ((abs_int + b_int).astype(f32) ** 2).astype(f32)
    y = (sign * (y + c_int).astype(f32)).astype(f32)                            # :212-213
    y = np.floor((y / f32(2 ** 6)).astype(f32))                                 # :215
    return (x_int * (y + shift_int).astype(f32)).astype(f32), s_out             # :231


def softmax_constants(s, act_min, act_max):
    """Scalar constants of IBERTIntSoftmax.forward (:250-319): x0_int, b_int, c_int, exp_sf, act_sf, (m, e) of the
    internal 16-bit QuantAct."""
    s = f32(s)
    x0 = -0.6931
    a, b, c = 0.35815147, 0.96963238 / 0.35815147, 1.0 / 0.35815147            # :261-263
    x0_int = np.floor(f32(f32(x0) / s))                                        # :287
    b_int = np.floor(f32(f32(b) / s))                                          # :277
    c_int = np.floor(f32(f32(c) / f32(s * s)))                                 # :278
    exp_sf = f32(f32(f32(a) * f32(s * s)) / f32(2 ** 30))                      # :282, 294
    act_sf = max(f32(f32(max(-f32(act_min), f32(act_max))) / f32(2 ** 15 - 1)), f32(EPS32))   # quant_utils.py:52-70, 16 bit
    m, e = _dyadic(exp_sf, act_sf)
    return f32(x0_int), f32(b_int), f32(c_int), exp_sf, act_sf, m, e


def softmax(k, s, act_min, act_max, output_bit=8, return_exp=False):
    """k [..., L] integers with scale s; (act_min, act_max) = range of the module's internal QuantAct(16).
    -> (out_int float32 in [0, 2^(output_bit-1)), s_out = 2 / 2^output_bit, n_inexact_rows)."""
    s = f32(s)
    n = 30
    x0_int, b_int, c_int, exp_sf, act_sf, m, e = softmax_constants(s, act_min, act_max)
    x_int = ((np.asarray(k).astype(f32) * s).astype(f32) / s).astype(f32)      # :303
    x_int = (x_int - x_int.max(axis=-1, keepdims=True)).astype(f32)            # :305-306
    x_int = np.maximum(x_int, f32(n * x0_int)).astype(f32)                     # :288
    q = np.floor((x_int / x0_int).astype(f32))                                 # :290
    r = (x_int - (x0_int * q).astype(f32)).astype(f32)                         # :291
    z = ((r * (r + b_int).astype(f32)).astype(f32) + c_int).astype(f32)        # :279-281
    ex = np.floor((z * np.exp2((n - q).astype(f32)).astype(f32)).astype(f32))  # :293
    ex = np.maximum(ex, f32(0))
    # internal QuantAct(16) = fixedpoint_mul(exp_int, exp_sf, 16) (quant_utils.py:220-245)
    z_int = np.rint((ex / exp_sf).astype(f32))
    q16 = np.clip(np.rint(z_int.astype(np.float64) * m / 2.0 ** e), -32768, 32767).astype(f32)
    exp_int = ((q16 * act_sf).astype(f32) / act_sf).astype(f32)                # :309-310
    exact = exp_int.astype(np.float64).sum(axis=-1, keepdims=True)
    ssum = exact.astype(f32)                                                   # :311 float32 row sum, exact below 2^24
    ninexact = int((exact >= 2 ** 24).sum())
    factor = np.floor((f32(2 ** 32) / ssum).astype(f32))                       # :313
    out = np.floor(((exp_int * factor).astype(f32) / f32(2 ** (32 - output_bit + 1))).astype(f32))   # :314
    res = (out, f32(2 / 2 ** output_bit), ninexact)
    return res + (ex,) if return_exp else res


def layernorm_constants(gamma, beta):
    """bias_int[C], s_out[C] of IBERTIntLayerNorm.forward (:147-156)."""
    gamma, beta = np.asarray(gamma, f32), np.asarray(beta, f32)
    C = gamma.shape[0]
    sf = f32(np.sqrt(f32(C)).astype(f32) / f32(2 ** 30))
    bias_int = np.floor(((beta / gamma).astype(f32) / sf).astype(f32))
    return bias_int.astype(f32), (sf * gamma).astype(f32)


def layernorm(k, s, gamma, beta, shift=0.0):
    """k [..., C] integers with scale s -> (y_int float32, s_out[C], n_inexact_rows).  :112-158 (use_int_sqrt False)"""
    s = f32(s)
    C = np.asarray(k).shape[-1]
    bias_int, s_out = layernorm_constants(gamma, beta)
    x_int = ((np.asarray(k).astype(f32) * s).astype(f32) / s).astype(f32)      # :126
    sx = x_int.astype(np.float64).sum(axis=-1, keepdims=True)
    mean_int = np.rint((sx.astype(f32) / f32(C)).astype(f32))                  # :127
    y_int = (x_int - mean_int).astype(f32)                                     # :128
    sh = f32(2.0 ** shift)
    y_sh = np.floor((y_int / sh).astype(f32))                                  # :129
    exact = (y_sh * y_sh).astype(f32).astype(np.float64).sum(axis=-1, keepdims=True)        # :130 float32 squares
    var_int = exact.astype(f32)                                                # :130-131, exact below 2^24
    ninexact = int(((exact >= 2 ** 24) | (np.abs(sx) >= 2 ** 24)).sum())
    with np.errstate(divide="ignore", invalid="ignore"):   # a constant row has std = 0: factor = inf, 0 * inf = NaN, as the reference
        std_int = (np.floor(np.sqrt(var_int).astype(f32)) * sh).astype(f32)    # :142
        factor = np.floor((f32(2 ** 31) / std_int).astype(f32))                # :143
        y = np.floor(((y_int * factor).astype(f32) / f32(2)).astype(f32))      # :144
    return (y + bias_int).astype(f32), s_out, ninexact                         # :151
