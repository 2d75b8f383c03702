"""nn.Module mirror of the reference's fake-quant operators, backed by the HIP kernels.

Same class names, constructor arguments, registered buffers (state_dict keys),
``fix()/unfix()`` protocol and ``forward(x, scale, ...) -> (y, scale)`` contract as
/root/reference/models/quantization_utils/quant_modules.py:131-511, so the reference's model
files and checkpoints apply unchanged.  Tensors crossing a module boundary are the reference's
float32 ``value = integer * scale`` views; inside, each module converts to integers, runs the
integer kernel through the C ABI (``ivit_*`` in include/ivit_hip.h) and converts back.  This is the
compatibility path (one launch per conversion: calibration, I-BERT operators, non-8-bit widths, Swin);
``VisionTransformer.forward`` of a frozen model takes the fused int8 engine (engine.py) instead, and a
frozen I-ViT model that is called module by module anyway carries int8 between the modules (lazy.py:
the ``forward`` of each class below hands over to it, ``_slow`` is the ordinary path).

Scales are read back to the host to derive the dyadic (m, e) pairs exactly as batch_frexp does
(quant_utils.py:151-175); that is what the reference itself does on every call (numpy + Decimal).
Valid in the regime the parity contract covers (power-of-two activation scales).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..prepare import LinearParams, dyadic, f32, sym_scale
from . import lazy


# ----------------------------------------------------------------------------- device helpers
def _st():
    return _lib.stream_ptr()


# ----------------------------------------------------------------------------- I/O-statistics collector
# quant_modules.py:17-125 of the reference: forward hooks on every sub-module record the extrema of each layer's input and
# output, as floats and in integer units (x / scale), and the shapes; scripts/inference.py:367-400 attaches them by default and
# writes io_stats_*.pkl / .csv (summarised by scripts/analyze_io_stats.py).  Same names, same record keys, same quirks (a layer
# with a per-channel scale raises inside the hook and is silently dropped, :79-81).  The extrema come from the HIP reduction
# (ivit_minmax_f32); min / max of x / s for a positive scalar s are the correctly rounded quotients of min / max of x
# (monotone), so the integer-unit columns equal the reference's without a pass over x / s.
# Hooks sit on the sub-modules: a model carrying them runs module by module (dispatch.takes_engine), never the fused engine.
_LAYER_IO_STATS = []        # global buffer
_IO_STATS_ENABLED = True    # global enable / disable flag


def _minmax(t: torch.Tensor):
    ta = t.detach().contiguous().float()
    mm = torch.empty(2, dtype=torch.float32, device=ta.device)
    _lib.call("ivit_minmax_f32", _lib.ptr(ta), ta.numel(), _lib.ptr(mm), _st())
    return mm[0], mm[1]


def _collect_io_stats(module, inputs, output, module_name):
    if not _IO_STATS_ENABLED:
        return
    try:
        x_scaled = inputs[0]
        scale_x = inputs[1] if len(inputs) > 1 and torch.is_tensor(inputs[1]) else None
        y_scaled = output[0] if isinstance(output, (tuple, list)) else output
        scale_y = output[1] if isinstance(output, (tuple, list)) and torch.is_tensor(output[1]) else None
        x_lo, x_hi = _minmax(x_scaled)
        y_lo, y_hi = _minmax(y_scaled)
        sx = scale_x.item() if scale_x is not None else None          # a per-channel scale raises here, as in the reference
        sy = scale_y.item() if scale_y is not None else None

        def ints(lo, hi, sc):      # (x / s).min(), (x / s).max() for s > 0; a negative s swaps the ends
            if sc is None:
                return None, None
            a, b = (lo / sc).item(), (hi / sc).item()
            return (a, b) if a <= b else (b, a)

        xi, yi = ints(x_lo, x_hi, scale_x), ints(y_lo, y_hi, scale_y)
        rec = {"layer": module_name, "type": module.__class__.__name__,
               "min_in": x_lo.item(), "max_in": x_hi.item(), "min_out": y_lo.item(), "max_out": y_hi.item(),
               "scale_in": sx, "scale_out": sy,
               "min_in_int": xi[0], "max_in_int": xi[1], "min_out_int": yi[0], "max_out_int": yi[1],
               "shape_in": tuple(x_scaled.shape), "shape_out": tuple(y_scaled.shape)}
        if isinstance(module, QuantMatMul):
            A, sA, B, sB = inputs[0], inputs[1], inputs[2], inputs[3]
            ai, bi = ints(*_minmax(A), sA), ints(*_minmax(B), sB)
            rec.update({"min_A_int": ai[0], "max_A_int": ai[1], "shape_A": tuple(A.shape),
                        "min_B_int": bi[0], "max_B_int": bi[1], "shape_B": tuple(B.shape)})
        _LAYER_IO_STATS.append(rec)
    except Exception:
        pass    # swallow any hook error so that evaluation never breaks (quant_modules.py:79-81)


def attach_io_stat_hooks(model: nn.Module):
    """Recursively attach `_collect_io_stats` as a forward hook to every sub-module (quant_modules.py:83-89)."""
    from functools import partial
    for name, module in model.named_modules():
        if module is model:      # skip the top-level container to avoid double-logging
            continue
        module.register_forward_hook(partial(_collect_io_stats, module_name=name))
    model._io_stat_hooks = True


def io_stats_enabled() -> bool:
    return _IO_STATS_ENABLED


def enable_io_stats():
    global _IO_STATS_ENABLED
    _IO_STATS_ENABLED = True


def disable_io_stats():
    global _IO_STATS_ENABLED
    _IO_STATS_ENABLED = False


def clear_io_stats():
    _LAYER_IO_STATS.clear()


def get_io_stats_df():
    """a new DataFrame of all gathered statistics (quant_modules.py:110-112)"""
    import pandas as pd
    return pd.DataFrame(_LAYER_IO_STATS)


def save_io_stats_df(path: str = "io_stats.pkl", to_csv: bool = False):
    """Export the collected statistics to `path` (pickle) plus an optional CSV next to it; returns the DataFrame (:115-125)."""
    df = get_io_stats_df()
    df.to_pickle(path)
    if to_csv:
        df.to_csv(path.rsplit(".", 1)[0] + ".csv", index=False)
    return df


from .quant_utils import *  # noqa: E402,F401,F403  (the reference's quant_modules does the same, quant_modules.py:15)


def _dev_table(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def to_int32(x: torch.Tensor, scale: torch.Tensor, trunc: bool = False) -> torch.Tensor:
    """z = round(x / scale) (quant_utils.py:220) or trunc (the `.to(int32)` of ivit_modules.py:38,107);
    scale has 1 or x.shape[-1] entries."""
    x = x.contiguous().float()
    s = scale.reshape(-1).contiguous().float()
    C = x.shape[-1]
    assert s.numel() in (1, C), "scale must be per tensor or per last-dim channel"
    z = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    _lib.call("ivit_f32_to_i32", _lib.ptr(x), x.numel() // C, C, _lib.ptr(s), s.numel(), int(trunc), _lib.ptr(z), _st())
    return z


def to_float(z: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    z = z.contiguous()
    s = scale.reshape(-1).contiguous().float()
    C = z.shape[-1]
    assert s.numel() in (1, C)
    y = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    _lib.call("ivit_i32_to_f32", _lib.ptr(z), z.numel() // C, C, _lib.ptr(s), s.numel(), _lib.ptr(y), _st())
    return y


def narrow_i8(z: torch.Tensor, what: str) -> torch.Tensor:
    out = torch.empty(z.shape, dtype=torch.int8, device=z.device)
    flag = torch.zeros(1, dtype=torch.int32, device=z.device)
    _lib.call("ivit_narrow_i32_i8", _lib.ptr(z), _lib.ptr(out), z.numel(), _lib.ptr(flag), _st())
    if int(flag.item()) != 0:
        raise _lib.IvitError(f"{what}: integer activations exceed 8 bits; the int8 MFMA kernels cannot represent them")
    return out


def _pad_k(k: int) -> int:
    """the GEMM kernels step K in 64-byte slabs; operands are zero-padded up to it (Swin: K = 96, 48)"""
    return (k + 63) // 64 * 64


def _pad_cols(a: np.ndarray, kp: int) -> np.ndarray:
    if a.shape[1] == kp:
        return a
    out = np.zeros((a.shape[0], kp), a.dtype)
    out[:, :a.shape[1]] = a
    return out


def _me_tables(pre_sf: torch.Tensor, z_sf, device):
    m, e = dyadic(pre_sf.detach().reshape(-1).cpu().numpy(), f32(z_sf))
    return _dev_table(m.view(np.int32), device), _dev_table(e, device), m.size


# ----------------------------------------------------------------------------- QuantLinear
class QuantLinear(nn.Linear):
    """quant_modules.py:131-226."""

    def __init__(self, in_features, out_features, bias=True, weight_bit=8, bias_bit=32, per_channel=True,
                 quant_mode="symmetric"):
        super().__init__(in_features, out_features, bias)
        if quant_mode != "symmetric":
            raise ValueError(f"unknown quant mode: {quant_mode}")
        if weight_bit != 8 or not per_channel:
            raise NotImplementedError("the HIP path implements 8-bit per-channel weights (the reference default)")
        self.weight_bit, self.bias_bit, self.per_channel, self.quant_mode = weight_bit, bias_bit, per_channel, quant_mode
        self.quantize_bias = bias_bit is not None
        self.register_buffer("fc_scaling_factor", torch.zeros(self.out_features))
        self.register_buffer("weight_integer", torch.zeros_like(self.weight))
        if self.bias is not None:
            self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._cache = None

    def __repr__(self):
        return "(" + super().__repr__() + f" weight_bit={self.weight_bit}, quant_mode={self.quant_mode})"

    def fix(self):
        pass

    def unfix(self):
        pass

    def _params(self, s_in: float):
        key = (self.weight._version, None if self.bias is None else self.bias._version, s_in, self.weight.device)
        if self._cache is None or self._cache[0] != key:
            lp = LinearParams(self.weight.detach().cpu().numpy(),
                              None if self.bias is None else self.bias.detach().cpu().numpy(), s_in)
            dev = self.weight.device
            W8h, b32h = _pad_cols(lp.W8, _pad_k(lp.K)), lp.b32
            if W8h.shape[0] % 4:          # the GEMM wants N % 4 == 0: zero rows, sliced off again in forward
                padn = -W8h.shape[0] % 4
                W8h = np.concatenate([W8h, np.zeros((padn, W8h.shape[1]), np.int8)])
                b32h = None if b32h is None else np.concatenate([b32h, np.zeros(padn, b32h.dtype)])
            W8 = _dev_table(W8h, dev)
            b32 = None if b32h is None else _dev_table(b32h, dev)
            s_acc = _dev_table(lp.s_acc, dev)
            self._publish(lp, dev)
            self._cache = (key, W8, b32, s_acc)
        return self._cache[1:]

    def _publish(self, lp, dev):
        """buffers the reference overwrites on every call (quant_modules.py:211-220)"""
        self.fc_scaling_factor = _dev_table(lp.sw, dev)
        self.weight_integer = _dev_table(lp.W8, dev).float().reshape(self.weight.shape)
        if lp.b32 is not None:
            self.bias_integer = _dev_table(lp.b32, dev).float()

    def forward(self, x, prev_act_scaling_factor=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(prev_act_scaling_factor)
            if x.q8 is not None and s_in is not None and s_in.size == 1 and self.in_features % 64 == 0:
                # int8-carrying input of a frozen model: nothing runs here; the QuantAct behind launches GEMM + requantisation
                c = lazy.linear_consts(self, s_in, x.device)
                return lazy.pending("linear", self, (*x.shape[:-1], self.out_features), x.device, (x,),
                                    (prev_act_scaling_factor,), c["s_acc"])
            x = x.to_float()
        return self._slow(x, prev_act_scaling_factor)

    def _slow(self, x, prev_act_scaling_factor=None):
        assert prev_act_scaling_factor is not None and prev_act_scaling_factor.shape == (1,)
        W8, b32, s_acc = self._params(float(prev_act_scaling_factor.item()))
        K, N = self.in_features, self.out_features
        if K % 16 != 0:
            raise NotImplementedError("ivit_gemm_i8_i32 needs in_features % 16 == 0")
        a8 = narrow_i8(to_int32(x, prev_act_scaling_factor), "QuantLinear input")
        M = a8.numel() // K
        Kp = _pad_k(K)
        if Kp != K:
            a8p = torch.zeros(M, Kp, dtype=torch.int8, device=x.device)
            a8p[:, :K] = a8.reshape(M, K)
            a8 = a8p
        Np = W8.shape[0]
        acc = torch.empty((*x.shape[:-1], Np), dtype=torch.int32, device=x.device)
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(a8), Kp, _lib.ptr(W8), Kp, _lib.ptr(b32), _lib.ptr(acc), Np, M, Np, Kp, _st())
        if Np != N:
            acc = acc[..., :N].contiguous()
        return to_float(acc, s_acc), s_acc


# ----------------------------------------------------------------------------- QuantAct
class QuantAct(nn.Module):
    """quant_modules.py:229-387 (fixedpoint_mul quant_utils.py:193-253; input mode :79-97)."""

    def __init__(self, activation_bit=8, act_range_momentum=0.95, running_stat=True, per_channel=False,
                 channel_len=None, quant_mode="symmetric"):
        super().__init__()
        if quant_mode != "symmetric":
            raise ValueError(f"unknown quant mode: {quant_mode}")
        if per_channel:
            raise NotImplementedError("per-channel activation ranges are not used by the ViT path")
        self.activation_bit, self.act_range_momentum = activation_bit, act_range_momentum
        self.running_stat, self.quant_mode, self.per_channel = running_stat, quant_mode, per_channel
        self.percentile = None
        self.register_buffer("x_min", torch.zeros(1))
        self.register_buffer("x_max", torch.zeros(1))
        self.register_buffer("act_scaling_factor", torch.zeros(1))

    def __repr__(self):
        return (f"{self.__class__.__name__}(activation_bit={self.activation_bit}, quant_mode: {self.quant_mode}, "
                f"Act_min: {self.x_min.item():.2f}, Act_max: {self.x_max.item():.2f})")

    def fix(self):
        self.running_stat = False

    def unfix(self):
        self.running_stat = True

    def _observe(self, x_act):
        """calibration statistics (quant_modules.py:310-360): min / max by the HIP reduction, then the reference's
        initialise / running-min-max / EMA update on the two scalars"""
        xa = x_act.detach().contiguous().float()
        mm = torch.empty(2, dtype=torch.float32, device=xa.device)
        _lib.call("ivit_minmax_f32", _lib.ptr(xa), xa.numel(), _lib.ptr(mm), _st())
        self._observe_update(mm[0], mm[1])

    def _observe_update(self, x_min, x_max):
        """the reference's update of its range by one observation (quant_modules.py:346-360)"""
        if torch.eq(self.x_min, self.x_max).all():
            self.x_min = self.x_min + x_min
            self.x_max = self.x_max + x_max
        elif self.act_range_momentum == -1:
            self.x_min = torch.min(self.x_min, x_min)
            self.x_max = torch.max(self.x_max, x_max)
        else:
            mo = self.act_range_momentum
            self.x_min = self.x_min * mo + x_min * (1 - mo)
            self.x_max = self.x_max * mo + x_max * (1 - mo)

    def _frozen_scale(self, device):
        """(s_out on the host, QS, plain scale tensor) of a fixed QuantAct, recomputed only when its range buffers change"""
        key = (id(self.x_min), self.x_min._version, id(self.x_max), self.x_max._version, str(device))
        c = self.__dict__.get("_frozen")
        if c is None or c[0] != key:
            s_out = sym_scale(float(self.x_min.reshape(-1)[0]), float(self.x_max.reshape(-1)[0]), self.activation_bit)
            qs = lazy.QS.make(s_out, device)
            c = self.__dict__["_frozen"] = (key, f32(s_out), qs, qs.as_subclass(torch.Tensor), (self.x_min, self.x_max))
        return c[1], c[2], c[3]

    def _fast(self, x, pre_sf, identity, identity_sf):
        """frozen 8-bit QuantAct of an int8-carrying forward (lazy.py): int8 out, one fused launch, nothing read back"""
        if pre_sf is None:
            if isinstance(x, lazy.QT) or identity is not None or not x.is_cuda:
                return None
            s_out, qs, plain = self._frozen_scale(x.device)
            xin = x.detach().contiguous().float()
            q8 = torch.empty(xin.shape, dtype=torch.int8, device=x.device)
            _lib.call("ivit_quantize_input_f32_i8", _lib.ptr(xin), _lib.ptr(q8), xin.numel(), float(f32(1.0) / s_out), _st())
            self.act_scaling_factor = plain
            return lazy.QT.wrap(q8.shape, x.device, q8=q8, scale=qs), qs
        if not isinstance(x, lazy.QT):
            return None
        s_out, qs, plain = self._frozen_scale(x.device)
        r = lazy.resolve(self, x, pre_sf, identity, identity_sf, s_out, qs)
        if r is None:
            return None
        self.act_scaling_factor = plain
        return r, qs

    def forward(self, x, pre_act_scaling_factor=None, identity=None, identity_scaling_factor=None,
                specified_min=None, specified_max=None):
        if (lazy.active() and not self.running_stat and self.activation_bit == 8 and specified_min is None
                and specified_max is None):
            r = self._fast(x, pre_act_scaling_factor, identity, identity_scaling_factor)
            if r is not None:
                return r
        if isinstance(x, lazy.QT):
            x = x.to_float()
        if isinstance(identity, lazy.QT):
            identity = identity.to_float()
        return self._slow(x, pre_act_scaling_factor, identity, identity_scaling_factor, specified_min, specified_max)

    def _slow(self, x, pre_act_scaling_factor=None, identity=None, identity_scaling_factor=None,
              specified_min=None, specified_max=None):
        identity_in = identity
        if self.running_stat:
            self._observe(x if identity is None else identity + x)
        x_min = self.x_min if specified_min is None else specified_min
        x_max = self.x_max if specified_max is None else specified_max
        s_out = sym_scale(float(x_min.reshape(-1)[0]), float(x_max.reshape(-1)[0]), self.activation_bit)
        self.act_scaling_factor = torch.full((1,), float(s_out), dtype=torch.float32, device=x.device)
        bits = self.activation_bit
        if pre_act_scaling_factor is None:
            # input mode: clamp(round(1/s * x))  (SymmetricQuantFunction)
            xin = x.contiguous().float()
            if bits == 8:
                q8 = torch.empty(xin.shape, dtype=torch.int8, device=x.device)
                _lib.call("ivit_quantize_input_f32_i8", _lib.ptr(xin), _lib.ptr(q8), xin.numel(),
                          float(f32(1.0) / s_out), _st())
                q = q8.to(torch.int32)
            else:      # e.g. pos_encoding_bw = 16 (vit_quant.py:181)
                q = torch.empty(xin.shape, dtype=torch.int32, device=x.device)
                _lib.call("ivit_quantize_input_f32_i32", _lib.ptr(xin), _lib.ptr(q), xin.numel(),
                          float(f32(1.0) / s_out), int(bits), _st())
        else:
            z = to_int32(x, pre_act_scaling_factor)
            C = x.shape[-1]
            if pre_act_scaling_factor.numel() == 1 and float(pre_act_scaling_factor.reshape(-1)[0]) < 0:
                # a negative incoming scale (the I-BERT GELU's, ibert_modules.py:213,232): round-half-even and the
                # half-away-from-zero mantissa rounding of batch_frexp are both odd functions, so
                # requant(z, s) == requant(-z, -s) exactly
                z = -z
                pre_act_scaling_factor = -pre_act_scaling_factor
            m, e, n_me = _me_tables(pre_act_scaling_factor, s_out, x.device)
            z2 = m2 = e2 = None
            n2 = 0
            if identity is not None:
                if identity.shape != x.shape:  # e.g. pos_embed [1,T,C] against [B,T,C] (vit_quant.py:296)
                    identity = identity.expand_as(x)
                z2 = to_int32(identity, identity_scaling_factor)
                m2, e2, n2 = _me_tables(identity_scaling_factor, s_out, x.device)
            q = torch.empty(z.shape, dtype=torch.int32, device=x.device)
            _lib.call("ivit_requant_i32", _lib.ptr(z), z.numel() // C, C, _lib.ptr(m), _lib.ptr(e), n_me,
                      _lib.ptr(z2), _lib.ptr(m2), _lib.ptr(e2), n2, bits, _lib.ptr(q), _st())
        y = to_float(q, self.act_scaling_factor)
        # Memory layout of the result, as torch gives it to the reference's chain of elementwise ops: the strides of a dense
        # permuted input are kept, and the two-operand form ends in `output1 + output` (quant_utils.py:245) with the IDENTITY's
        # term first, whose layout wins.  A consumer whose float32 reduction order depends on the layout (IVITIntLayerNorm behind
        # the Swin patch embedding's transpose, layers_quant.py:198-201, and every LayerNorm of Swin's first stage) must see it.
        lay = identity_in if (identity_in is not None and identity_in.shape == x.shape) else x
        if not lay.is_contiguous() and y.shape == lay.shape:
            y = torch.empty_like(lay, dtype=torch.float32).copy_(y)
        return y, self.act_scaling_factor


# ----------------------------------------------------------------------------- QuantMatMul
class QuantMatMul(nn.Module):
    """quant_modules.py:390-409: (A/sA) @ (B/sB) with scale sA*sB."""

    def __init__(self):
        super().__init__()
        self.register_buffer("act_scaling_factor", torch.zeros(1))

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, A, pre_act_scaling_factor_A, B, pre_act_scaling_factor_B):
        if isinstance(A, lazy.QT) or isinstance(B, lazy.QT):
            sa, sb = lazy.host_of(pre_act_scaling_factor_A), lazy.host_of(pre_act_scaling_factor_B)
            if (isinstance(A, lazy.QT) and isinstance(B, lazy.QT) and sa is not None and sb is not None and sa.size == 1
                    and sb.size == 1 and A.shape[:-2] == B.shape[:-2] and A.shape[-1] == B.shape[-2]):
                s = pre_act_scaling_factor_A * pre_act_scaling_factor_B
                self.act_scaling_factor = s.as_subclass(torch.Tensor)
                return lazy.pending("matmul", self, (*A.shape[:-1], B.shape[-1]), A.device, (A, B),
                                    (pre_act_scaling_factor_A, pre_act_scaling_factor_B), s)
            A = A.to_float() if isinstance(A, lazy.QT) else A
            B = B.to_float() if isinstance(B, lazy.QT) else B
        return self._slow(A, B, pre_act_scaling_factor_A, pre_act_scaling_factor_B)

    def _slow(self, A, B, pre_act_scaling_factor_A, pre_act_scaling_factor_B):
        a32 = to_int32(A, pre_act_scaling_factor_A)
        b8 = narrow_i8(to_int32(B, pre_act_scaling_factor_B), "QuantMatMul B")  # to_int32 makes B contiguous [.., K, N]
        Tq, Kd = A.shape[-2], A.shape[-1]
        N = B.shape[-1]
        assert B.shape[-2] == Kd and A.shape[:-2] == B.shape[:-2]
        batch = a32.numel() // (Tq * Kd)
        out = torch.empty((*A.shape[:-1], N), dtype=torch.int32, device=A.device)
        amax = int(a32.abs().max())
        if amax <= 127:
            a8 = a32.to(torch.int8)
            _lib.call("ivit_bgemm_pv_i8", _lib.ptr(a8), _lib.ptr(b8), _lib.ptr(out), batch, Tq, Kd, N, _st())
        elif amax <= 32767:      # Shiftmax with softmax_bw = 16 feeds P . V with 16-bit probabilities
            a16 = a32.to(torch.int16)
            _lib.call("ivit_bgemm_pv_i16_i8", _lib.ptr(a16), _lib.ptr(b8), _lib.ptr(out), batch, Tq, Kd, N, _st())
        else:                    # I-BERT's softmax at 16 bits reaches 2^15 (a one-hot row); the entry point bounds the int32 sum
            _lib.call("ivit_bgemm_pv_i32_i8", _lib.ptr(a32), _lib.ptr(b8), _lib.ptr(out), batch, Tq, Kd, N, amax, _st())
        s = (pre_act_scaling_factor_A * pre_act_scaling_factor_B).float()
        self.act_scaling_factor = s
        return to_float(out, s), s


# ----------------------------------------------------------------------------- QuantConv2d
class QuantConv2d(nn.Conv2d):
    """quant_modules.py:412-511, for the patch-embedding geometry (kernel == stride, no padding)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 weight_bit=8, bias_bit=32, quant_mode="symmetric", per_channel=True, weight_percentile=0):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias)
        if quant_mode != "symmetric" or not per_channel or weight_bit != 8:
            raise NotImplementedError("the HIP path implements symmetric 8-bit per-channel conv weights")
        self.weight_bit, self.bias_bit, self.quant_mode, self.per_channel = weight_bit, bias_bit, quant_mode, per_channel
        self.weight_percentile = weight_percentile
        self.quantize_bias = bias_bit is not None
        self.register_buffer("conv_scaling_factor", torch.zeros(self.out_channels))
        self.register_buffer("weight_integer", torch.zeros_like(self.weight))
        self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._cache = None

    def __repr__(self):
        return "(" + super().__repr__() + f" weight_bit={self.weight_bit}, quant_mode={self.quant_mode})"

    def fix(self):
        pass

    def unfix(self):
        pass

    def _publish(self, lp, dev):
        self.conv_scaling_factor = _dev_table(lp.sw, dev)
        self.weight_integer = _dev_table(lp.W8, dev).float().reshape(self.weight.shape)
        self.bias_integer = _dev_table(lp.b32, dev).float()

    def forward(self, x, pre_act_scaling_factor=None):
        if isinstance(x, lazy.QT):
            s_in = lazy.host_of(pre_act_scaling_factor)
            kh, kw = self.kernel_size
            K = self.in_channels * kh * kw
            if (x.q8 is not None and s_in is not None and s_in.size == 1 and K % 64 == 0 and self.groups == 1
                    and self.dilation == (1, 1) and self.bias is not None and x.dim() == 4
                    and kh == kw == self.stride[0] == self.stride[1] and self.padding == (0, 0)
                    and x.shape[2] % kh == 0 and x.shape[3] % kw == 0):
                c = lazy.linear_consts(self, s_in, x.device)
                shape = (x.shape[0], self.out_channels, x.shape[2] // kh, x.shape[3] // kw)
                return lazy.pending("conv", self, shape, x.device, (x,), (pre_act_scaling_factor,), c["s_acc"].view(1, -1, 1, 1))
            x = x.to_float()
        return self._slow(x, pre_act_scaling_factor)

    def _slow(self, x, pre_act_scaling_factor=None):
        kh, kw = self.kernel_size
        if not (kh == kw == self.stride[0] == self.stride[1] and self.padding == (0, 0) and self.groups == 1
                and self.dilation == (1, 1)):
            raise NotImplementedError("QuantConv2d on the HIP path is the non-overlapping patch convolution")
        B, Cin, H, Wd = x.shape
        assert H == Wd and H % kh == 0
        s_in = float(pre_act_scaling_factor.reshape(-1)[0])
        key = (self.weight._version, self.bias._version, s_in, self.weight.device)
        if self._cache is None or self._cache[0] != key:
            lp = LinearParams(self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(), s_in)
            dev = x.device
            self._cache = (key, _dev_table(_pad_cols(lp.W8, _pad_k(lp.K)), dev), _dev_table(lp.b32, dev),
                           _dev_table(lp.s_acc, dev))
            self._publish(lp, dev)
        _, W8, b32, s_acc = self._cache
        g = H // kh
        K = _pad_k(Cin * kh * kw)
        A = torch.zeros(B * g * g, K, dtype=torch.int8, device=x.device)
        xin = x.contiguous().float()
        _lib.call("ivit_quantize_patchify_ld_f32_i8", _lib.ptr(xin), _lib.ptr(A), K, B, Cin, H, kh,
                  float(f32(1.0) / f32(s_in)), _st())
        N = self.out_channels
        acc = torch.empty(B * g * g, N, dtype=torch.int32, device=x.device)
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(A), K, _lib.ptr(W8), K, _lib.ptr(b32), _lib.ptr(acc), N, B * g * g, N, K,
                  _st())
        y = to_float(acc, s_acc).reshape(B, g, g, N).permute(0, 3, 1, 2).contiguous()
        return y, s_acc.view(1, -1, 1, 1)
