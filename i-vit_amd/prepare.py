"""Load-time (host) preparation of the integer parameters of the I-ViT path.

The reference recomputes all of this inside every forward call
(/root/reference/models/quantization_utils/quant_modules.py:202-220, 371-372, 486-504;
quant_utils.py:151-175, 221-228); they are constants of a frozen model, so the
MI355X path derives them once on the host, in the same float32 / float64
arithmetic, and keeps only integers and dyadic (m, e) pairs on the device.

numpy float32 scalars/arrays are IEEE single: `a * b`, `a / b`, np.rint,
np.floor, np.sqrt round exactly like the torch CPU ops they restate.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
EPS32 = np.finfo(np.float32).eps


def sym_scale(x_min, x_max, bits: int = 8) -> np.float32:
    """symmetric_linear_quantization_params, quant_utils.py:52-70."""
    n = f32(2 ** (bits - 1) - 1)
    mx = max(-f32(x_min), f32(x_max))
    return max(f32(f32(mx) / n), f32(EPS32))


def weight_scale(W2d: np.ndarray, bits: int = 8) -> np.ndarray:
    """Per-output-row weight scale, quant_modules.py:204-212 / 488-498."""
    W2d = np.asarray(W2d, dtype=f32)
    n = f32(2 ** (bits - 1) - 1)
    mx = np.maximum(-W2d.min(axis=1), W2d.max(axis=1)).astype(f32)
    return np.maximum((mx / n).astype(f32), f32(EPS32))


def quant_sym(x: np.ndarray, scale, bits: int = 8) -> np.ndarray:
    """SymmetricQuantFunction.forward, quant_utils.py:79-97: clamp(round(1./scale * x), -n-1, n).
    `scale` broadcasts against x (per-row weights: pass scale[:, None])."""
    x = np.asarray(x, dtype=f32)
    rs = (f32(1.0) / np.asarray(scale, dtype=f32)).astype(f32)
    q = np.rint((rs * x).astype(f32))
    lo, hi = f32(-(2.0 ** (bits - 1))), f32(2.0 ** (bits - 1) - 1)  # hi rounds to 2^31 for 32 bit, as in torch
    q = np.clip(q, lo, hi)
    return np.minimum(q.astype(np.float64), 2.0 ** 31 - 1).astype(np.int64).astype(np.int32)


def dyadic(pre_sf, z_sf):
    """(m, e) of fixedpoint_mul / batch_frexp (quant_utils.py:151-175, 221-228):
    new_scale = double(pre_sf)/double(float(z_sf)) = mant * 2^exp; m = round_half_up(mant*2^31); e = 31-exp."""
    pre = np.atleast_1d(np.asarray(pre_sf, dtype=f32)).astype(np.float64)
    ns = pre / np.float64(f32(z_sf))
    mant, ex = np.frexp(ns)
    m = np.floor(mant * 2.0 ** 31 + 0.5)
    assert np.all(m >= 2 ** 30) and np.all(m <= 2 ** 31), "degenerate requant ratio"
    return m.astype(np.uint32), (31 - ex).astype(np.int32)


def requant_host(z, m, e) -> np.ndarray:
    """RNE(z*m/2^e) in float64 exactly as quant_utils.py:229-230 (used for load-time constants)."""
    return np.rint(np.asarray(z, np.float64) * np.asarray(m, np.float64) / np.exp2(np.asarray(e, np.float64)))


class LinearParams:
    """Integer weights of a QuantLinear / QuantConv2d given the (fixed) input scale."""

    def __init__(self, W: np.ndarray, b, s_in):
        W2 = np.asarray(W, dtype=f32).reshape(W.shape[0], -1)
        self.sw = weight_scale(W2)
        self.W8 = quant_sym(W2, self.sw[:, None], 8).astype(np.int8)
        self.s_acc = (self.sw * f32(s_in)).astype(f32)  # bias_scaling_factor, quant_modules.py:217
        self.b32 = None if b is None else quant_sym(np.asarray(b, f32), self.s_acc, 32)
        self.K = W2.shape[1]

    def requant_to(self, s_out):
        m, e = dyadic(self.s_acc, s_out)
        if np.any(e < 31):
            raise ValueError("GEMM requantiser with multiplier > 1 (e < 31) is outside the kernels' contract")
        return m, e


class LayerNormParams:
    """Constants of IVITIntLayerNorm (ivit_modules.py:31-33, 53-62) + the QuantAct behind it."""

    def __init__(self, gamma, beta, s_out):
        gamma = np.asarray(gamma, f32)
        beta = np.asarray(beta, f32)
        C = gamma.shape[0]
        dim_sqrt = np.sqrt(f32(C)).astype(f32)
        sf = f32(dim_sqrt / f32(2 ** 30))
        self.bias_int = np.floor(((beta / gamma).astype(f32) / sf).astype(f32)).astype(f32)
        self.s_ln = (sf * gamma).astype(f32)
        self.m, self.e = dyadic(self.s_ln, s_out)
        if np.any(self.e < 40):
            # |z| < 2^31 after I-LayerNorm, so M <= 2^-9 keeps |z*M| < 2^22 (the kernel's float32 certificate range)
            raise ValueError("LayerNorm requantiser with multiplier > 2^-9 (e < 40) is outside the kernel's contract")


def pad_head(W8: np.ndarray, b32: np.ndarray, s_acc: np.ndarray):
    """The classifier GEMM (`ivit_gemm_i8_i32`) wants N % 4 == 0.  Any other class count is padded with classes that can
    never win the arg-max: zero weights, bias -2^31, scale 1 (logit -2.1e9).  -> (W8, b32, s_acc, N_true)."""
    N = W8.shape[0]
    pad = -N % 4
    if pad:
        W8 = np.concatenate([W8, np.zeros((pad, W8.shape[1]), W8.dtype)])
        b32 = np.concatenate([b32, np.full(pad, -2 ** 31, b32.dtype)])
        s_acc = np.concatenate([s_acc, np.ones(pad, s_acc.dtype)])
    return W8, b32, s_acc, N


# ---------------------------------------------------------------------------------------------------------------------
# Natural ("as calibrated") activation scales.  A QuantAct hands its consumer fl(q*s) (quant_modules.py:387) and the
# consumer divides by s again (ivit_modules.py:36, 106, 165): it sees phi_s(q) = fl(fl(q*s)/s), a float32 within an ulp or
# two of q.  round() sites recover q; I-LayerNorm (mean over phi, then trunc), ShiftGELU (trunc) and Shiftmax (phi itself
# through the whole float32 sequence) do not.  For s = 2^p phi is the identity and none of this is needed.
def phi_table(s) -> np.ndarray:
    q = np.arange(-128, 128, dtype=f32)
    s = f32(s)
    return ((q * s).astype(f32) / s).astype(f32)


def phi_is_identity(s, bits: int = 8) -> bool:
    """True when fl(fl(q*s)/s) == q for every `bits`-bit integer q (always for s = 2^p)."""
    if bits == 8:
        return bool(np.array_equal(phi_table(s), np.arange(-128, 128, dtype=f32)))
    q = np.arange(-(2 ** (bits - 1)), 2 ** (bits - 1), dtype=f32)
    return bool(np.array_equal(((q * f32(s)).astype(f32) / f32(s)).astype(f32), q))


def phi_tables(s):
    """-> (remap int8[256] = trunc(phi), phi float32[256]) or None when phi is the identity"""
    ph = phi_table(s)
    if np.array_equal(ph, np.arange(-128, 128, dtype=f32)):
        return None
    remap = np.trunc(ph).astype(np.int32)
    assert remap.min() >= -128 and remap.max() <= 127 and np.all(np.diff(remap) >= 0)
    return remap.astype(np.int8), ph


def shiftexp2d(s, n: int = 15) -> np.ndarray:
    """Shiftmax's exp_int as a function of (row max q, q) at a natural input scale s: the float32 sequence of
    ivit_modules.py:150-170 evaluated on phi(q) - phi(qmax), step by step in numpy float32 (each operation rounds exactly
    like the torch CPU op it restates).  uint32 [256, 256], entry [qmax+128, q+128]; entries with q > qmax are unused."""
    ph = phi_table(s)
    with np.errstate(over="ignore"):          # the unused q > qmax half overflows in 2^(n-q); it is zeroed below
        return _shiftexp2d(ph, f32(s), n)


def _shiftexp2d(ph, s, n, cols=None):
    """rows: the row maximum phi(qmax); columns: phi(q), or `cols` (the values of scores under Swin's shift mask: every column
    is then a valid pair, nothing is zeroed)"""
    d = ((ph if cols is None else cols)[None, :] - ph[:, None]).astype(f32)   # :168  x_int - x_int_max
    x = ((d + np.floor((d / f32(2)).astype(f32))).astype(f32) - np.floor((d / f32(16)).astype(f32))).astype(f32)   # :151
    x0 = np.floor(-(f32(1.0) / f32(s)))                                # :154  floor(-1.0 / s)
    x = np.maximum(x, f32(f32(n) * x0))                                # :155
    qq = np.floor((x / x0).astype(f32))                                # :157
    r = (x - (x0 * qq).astype(f32)).astype(f32)                        # :158
    ex = ((r / f32(2)).astype(f32) - x0).astype(f32)                   # :159
    ex = np.floor((ex * np.ldexp(f32(1.0), (f32(n) - qq).astype(np.int32)).astype(f32)).astype(f32))   # :160
    ex = np.maximum(ex, f32(0))
    if cols is None:
        ex = np.where(np.arange(256)[None, :] <= np.arange(256)[:, None], ex, f32(0))
    assert np.isfinite(ex).all() and ex.max() < 2.0 ** 32
    return ex.astype(np.uint32)


def shiftexp_band(tab2d: np.ndarray):
    """Band form of shiftexp2d for the LDS path of the attention kernel: band[qmax+128, j] = tab2d[qmax+128, qmax+128-j].
    -> (band uint32 [256, W], W) with W the smallest multiple of 16 such that every entry at distance >= W - 1 from the row
    maximum equals the saturated value (the exponent's argument is clamped at n*x0 there), or (None, 0) if W would exceed 256."""
    idx = np.arange(256)
    dist = idx[:, None] - idx[None, :]                  # qmax - q
    valid = dist >= 0
    sat = int(tab2d[255, 0])                            # distance 255: certainly clamped
    unsat = valid & (tab2d != sat)
    last = int(dist[unsat].max()) if unsat.any() else 0  # largest distance whose entry is not the saturated value
    W = -(-(last + 2) // 16) * 16                       # entry W - 1 must itself be saturated
    if W > 256:
        return None, 0
    j = np.arange(W)
    q = idx[:, None] - j[None, :]
    band = np.where(q >= 0, tab2d[idx[:, None], np.clip(q, 0, 255)], np.uint32(sat)).astype(np.uint32)
    assert np.all(band[:, W - 1] == sat)
    return np.ascontiguousarray(band), W


def window_shiftexp_band(s, masked: bool):
    """Band table for the Swin window-attention kernel at a natural Shiftmax input scale s (ivit_window_attention_i8_band), or
    (None, 0) when the literal float sequence has to run.  -> (band uint32 [256, W] or [1, W], W): one row when every valid
    entry of a column (q = qmax - j >= -128) has the same value, i.e. exp_int depends on the distance to the maximum alone (true
    for many scales: the float32 floors flip for few (qmax, q) pairs or none); band.shape[0] is the entry point's band_rows.  Unmasked scores see phi(q) = fl(fl(q*s)/s); scores under the shift
    mask see phi_m(q) = fl(fl(fl(q*s) - 100)/s) (swin_quant.py:151-156, ivit_modules.py:165).  The table form needs, when a mask
    is present: (a) no masked value can be the row maximum -- every query attends to itself unmasked, so max(phi_m) < min(phi)
    is enough; (b) every masked score's exp_int is the saturated value whatever the (unmasked) maximum: checked on all 256 x 256
    pairs with the same float32 sequence that builds the table."""
    s = f32(s)
    tab = shiftexp2d(s)
    band, W = shiftexp_band(tab)
    if band is None or W > 192:            # the kernel keeps 16 band rows per wave in LDS: widths up to 192
        return None, 0
    if masked:
        ph = phi_table(s)
        qv = np.arange(-128, 128, dtype=f32)
        phm = ((((qv * s).astype(f32) + f32(-100.0)).astype(f32)) / s).astype(f32)
        if not phm.max() < ph.min():
            return None, 0
        with np.errstate(over="ignore", invalid="ignore"):
            em = _shiftexp2d(ph, s, 15, cols=phm)
        if not np.all(em == tab[255, 0]):
            return None, 0
    valid = (np.arange(256)[:, None] - np.arange(W)[None, :]) >= 0
    if np.all((band == band[255][None, :]) | ~valid):
        return np.ascontiguousarray(band[255:256]), W
    return band, W


def markstein_division_ok(s, bits: int = 16) -> bool:
    """Is the 3-instruction quotient by the invariant s -- q0 = fl(x*r), e = fma(-s, q0, x), fma(e, r, q0), r = fl(1/s) -- the
    correctly rounded fl(x / s) for EVERY x = fl(q*s), q a `bits`-bit integer?  (Markstein's theorem says yes unless the
    significand of s is all ones; this checks the actual inputs exhaustively, emulating the two fmas in float64: both are an
    exact product plus one addend, rounded once.)  Enables the tiled natural-scale LayerNorm of the Swin engine."""
    s = f32(s)
    q = np.arange(-(2 ** (bits - 1)), 2 ** (bits - 1), dtype=f32)
    x = (q * s).astype(f32)
    ref = (x / s).astype(f32)
    r = f32(f32(1.0) / s)
    q0 = (x * r).astype(f32)
    e = (x.astype(np.float64) - np.float64(s) * q0.astype(np.float64)).astype(f32)          # exact difference, one rounding
    got = (q0.astype(np.float64) + e.astype(np.float64) * np.float64(r))
    # the float64 sum above can itself round (q0 is ~2^15, e*r ~2^-10): accept only when rounding it to float32 is unambiguous
    got32 = got.astype(f32)
    near_mid = np.abs(got - (got32.astype(np.float64) + np.spacing(got32).astype(np.float64) * np.where(got > got32, 0.5, -0.5))) \
        < np.abs(got) * 2.0 ** -50
    return bool(np.array_equal(got32, ref) and not near_mid.any())


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def input_lut_u8(s_in, mean=IMAGENET_MEAN, std=IMAGENET_STD, bits: int = 8) -> np.ndarray:
    """int8 [len(mean), 256]: what the float pipeline in front of the model makes of pixel value v of channel c -- torchvision's
    ToTensor (v / 255 in float32), Normalize ((x - mean) / std in float32) and the input QuantAct
    (SymmetricQuantFunction, quant_utils.py:79-97: clamp(round(1 / s * x))) -- each step in float32, in that order."""
    v = np.arange(256, dtype=f32) / f32(255.0)
    n = 2 ** (bits - 1) - 1
    inv = f32(1.0) / f32(s_in)
    rows = []
    for m, sd in zip(mean, std):
        x = ((v - f32(m)).astype(f32) / f32(sd)).astype(f32)
        rows.append(np.clip(np.rint((inv * x).astype(f32)), -n - 1, n).astype(np.int8))
    return np.stack(rows)
