"""The frozen module-by-module path without float round trips: int8-carrying tensors and fusion at the QuantAct.

The reference's model files (/root/reference/models/vit_quant.py:61-90, 142-155, 285-312) call QuantLinear, QuantAct,
IVITIntLayerNorm, ... one by one and move float32 `value = integer * scale` tensors between them.  Executed literally that is
a float -> integer conversion, an integer kernel and an integer -> float conversion per module, plus host read-backs of every
scale: 250 ms for a DeiT-B batch of 256 that the fused engine does in 6.5 ms.  This module keeps the calls and removes the
round trips, for a FROZEN model (every QuantAct fixed):

  * a frozen 8-bit QuantAct returns a `QT` -- a torch.Tensor subclass with the float tensor's shape / dtype / device but an
    **int8 payload** (no float storage) -- and a `QS` scale tensor that carries its value on the host as well;
  * a module whose input is a QT does not compute: it returns a QT holding a **pending node** (linear, conv, layer norm, GELU,
    the matmul -> scale -> QuantAct -> softmax -> matmul chain of attention);
  * the shape operations the model files apply in between (reshape, permute, transpose, indexing, unbind, flatten, eval-mode
    dropout, `* scalar`) act on the payload, or are recorded on the pending node;
  * the NEXT QuantAct launches ONE fused integer kernel for the node (GEMM + requantisation, LayerNorm + requantisation, GELU
    table, fused attention, residual add) -- the kernels of the fused engine -- and returns a QT again;
  * anything else that touches a QT (an unexpected torch function, a hook, the caller reading the logits) materialises the
    float tensor the reference would have produced, through the module's ordinary path, and continues from there.

Every (m, e) pair, table and integer weight is derived on the host from host-side scales and cached per module, so after the
first (warm-up) forward a frozen forward reads nothing back from the device: it runs under
`torch.cuda.set_sync_debug_mode("error")` and can be captured into a HIP graph (tests/test_gpu_modules.py).
Covers the I-ViT and the I-BERT operator families (ivit_modules.py, ibert_modules.py) at 8-bit QuantAct widths; other configurations
(16-bit widths, mixed families, Swin) take the ordinary module path through the materialisation rule above.
"""
from __future__ import annotations

import os

import threading
import warnings

import numpy as np
import torch

from .. import _lib
from ..prepare import (LayerNormParams, LinearParams, dyadic, f32, phi_tables, quant_sym, shiftexp2d, shiftexp_band,
                       sym_scale)

ENABLED = os.environ.get("IVIT_LAZY", "1") != "0"
_ALWAYS = [os.environ.get("IVIT_LAZY") == "always"]      # enable_everywhere(): process-wide, independent of any open scope
_TLS = threading.local()                                  # .depth: open `scope(True)` blocks of THIS thread; .mat: nested to_float calls
STATS = {"fused": 0, "materialised": 0}     # fused launches at a QuantAct / float tensors materialised, since the last reset
_WARNED = set()


def active() -> bool:
    """inside `with lazy.scope(True):` (this thread) -- the model mirror opens it around the module-by-module forward of a frozen
    I-ViT model -- or after enable_everywhere()"""
    return ENABLED and (_ALWAYS[0] or getattr(_TLS, "depth", 0) > 0)


def enable_everywhere(on: bool = True):
    """Frozen 8-bit QuantActs carry int8 wherever they are called from -- for callers that drive the modules themselves, e.g. the
    reference's own models/vit_quant.py imported on top of this package.  The mirror's VisionTransformer.forward
    opens the scope by itself; nothing else does by default (Swin's LayerNorms depend on the memory layout of their float inputs,
    DESIGN.md section 2, which an int8 payload does not carry).  Same as IVIT_LAZY=always in the environment."""
    _ALWAYS[0] = bool(on)       # never touches the scope depth: switching it off inside an open scope leaves the scope intact


class scope:
    def __init__(self, on: bool):
        self.on = bool(on) and ENABLED

    def __enter__(self):
        _TLS.depth = getattr(_TLS, "depth", 0) + int(self.on)

    def __exit__(self, *exc):
        _TLS.depth = getattr(_TLS, "depth", 0) - int(self.on)
        return False


def _st():
    return _lib.stream_ptr()


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


# ----------------------------------------------------------------------------------------------------------- scale tensors
class QS(torch.Tensor):
    """a scale tensor (real float32 storage on the device) that also knows its value on the host (`.host`, float32 array)"""

    @staticmethod
    def make(host, device):
        host = np.atleast_1d(np.asarray(host, dtype=f32)).copy()
        r = torch.Tensor._make_subclass(QS, torch.from_numpy(host.copy()).to(device))
        r.host = host
        r._mul = {}
        return r

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name in ("mul", "__mul__", "__rmul__") and len(args) == 2:
            a, b = (args[0], args[1]) if isinstance(args[0], QS) else (args[1], args[0])
            if isinstance(a, QS) and isinstance(b, (int, float)) and not isinstance(b, bool):
                key = float(b)
                if key not in a._mul:        # float32 tensor * python scalar: the scalar is taken at float32
                    a._mul[key] = QS.make((a.host * f32(key)).astype(f32), a.device)
                return a._mul[key]
            if isinstance(a, QS) and isinstance(b, QS) and a.host.size == 1 and b.host.size == 1:
                key = ("qs", id(b))
                if key not in a._mul:
                    a._mul[key] = QS.make((a.host * b.host).astype(f32), a.device)
                    a._mul[key]._keep = b      # keeps id(b) valid
                return a._mul[key]
        if name in ("view", "reshape") and isinstance(args[0], QS):
            with torch._C.DisableTorchFunctionSubclass():
                t = func(*args, **kwargs)
            r = torch.Tensor._make_subclass(QS, t)
            r.host = args[0].host
            r._mul = args[0]._mul
            return r
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if isinstance(out, QS):                 # any other op: a plain tensor, the host value is not carried on
            out = out.as_subclass(torch.Tensor)
        return out


def host_of(s):
    """host value of a scale tensor, or None if it is not a QS (the caller then takes the ordinary path)"""
    return s.host if isinstance(s, QS) else None


# ----------------------------------------------------------------------------------------------------------- int-carrying tensors
_VIEW = {"reshape", "view", "permute", "transpose", "__getitem__", "flatten", "contiguous", "unbind", "squeeze", "unsqueeze",
         "select", "narrow", "expand"}
_META = {"size", "dim", "numel", "__len__", "is_floating_point", "element_size", "is_contiguous", "stride", "storage_offset",
         "ndimension", "nelement", "is_complex", "get_device", "type"}


class QT(torch.Tensor):
    """float32-shaped tensor WITHOUT float storage: `q8` holds the int8 payload (a real tensor, same logical shape), or `node`
    a pending operation plus `views`, the shape operations recorded since"""

    @staticmethod
    def wrap(shape, device, q8=None, scale=None, node=None, views=()):
        r = torch.Tensor._make_wrapper_subclass(QT, tuple(shape), dtype=torch.float32, device=device, requires_grad=False)
        r._q8, r.scale, r.node, r.views = q8, scale, node, tuple(views)
        return r

    @property
    def q8(self):
        """the int8 payload; a deferred requantising GEMM (Requant) is launched the first time anybody asks for it"""
        if self._q8 is None and isinstance(self.node, Requant):
            self._q8 = self.apply_views(self.node.force())
            self.node, self.views = None, ()
        return self._q8

    # -- materialisation: the float tensor the reference's module would have returned
    def to_float(self, boundary=False):
        """`boundary`: the model hands its result to the caller (the one materialisation a forward is meant to have).  Any other
        outermost call means something inside the model looked at a float tensor -- correct, but that module then runs the float
        round trips the int8-carrying path exists to avoid: one warning per kind of producer (and lazy.STATS counts them all)."""
        STATS["materialised"] += 1
        depth = getattr(_TLS, "mat", 0)
        if depth == 0 and not boundary:
            what = "int8 payload" if self._q8 is not None else type(self.node).__name__ + (
                f"({self.node.kind})" if hasattr(self.node, "kind") else "")
            if what not in _WARNED:
                _WARNED.add(what)
                warnings.warn(f"ivit_amd.lazy: an int8-carrying activation ({what}) was materialised as float32 inside the model; the "
                              "modules behind it run their float form (about 30x slower per module) -- see lazy.STATS", RuntimeWarning,
                              stacklevel=2)
        _TLS.mat = depth + 1
        try:
            if self.q8 is not None:
                return self.q8.to(torch.float32) * self.scale.as_subclass(torch.Tensor).reshape(-1)[0]
            t = self.node.to_float()
            for fn in self.views:
                t = fn(t)
            return t
        finally:
            _TLS.mat = depth

    def apply_views(self, t):
        for fn in self.views:
            t = fn(t)
        return t

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by an operator that slipped past __torch_function__: materialise and run it on real tensors
        def real(t):
            return t.to_float() if isinstance(t, QT) else t
        return func(*torch.utils._pytree.tree_map(real, args), **torch.utils._pytree.tree_map(real, kwargs or {}))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        self = args[0] if args and isinstance(args[0], QT) else None
        if name == "__get__" or name in _META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if self is not None and name in _VIEW:
            rest = args[1:]

            def fn(t, _f=func, _r=rest, _k=kwargs):
                return _f(t, *_r, **_k)

            if self._q8 is not None:
                out = fn(self._q8)
                if isinstance(out, (tuple, list)):
                    return tuple(QT.wrap(o.shape, o.device, q8=o, scale=self.scale) for o in out)
                return QT.wrap(out.shape, out.device, q8=out, scale=self.scale)
            meta = fn(torch.empty(self.shape, device="meta"))
            if isinstance(meta, (tuple, list)):
                return tuple(QT.wrap(m.shape, self.device, scale=self.scale, node=self.node,
                                     views=self.views + ((lambda t, _fn=fn, _i=i: _fn(t)[_i]),)) for i, m in enumerate(meta))
            return QT.wrap(meta.shape, self.device, scale=self.scale, node=self.node, views=self.views + (fn,))
        if name in ("dropout", "dropout_", "feature_dropout", "alpha_dropout") and self is not None:
            training = kwargs.get("training", args[2] if len(args) > 2 else kwargs.get("train", True))
            if not training:
                return self
        if name in ("mul", "__mul__", "__rmul__") and len(args) == 2:
            a, b = (args[0], args[1]) if isinstance(args[0], QT) else (args[1], args[0])
            if isinstance(b, (int, float)) and not isinstance(b, bool):
                # value * c with the scale multiplied by c alongside (vit_quant.py:74-75): the integers stay what they are
                if a.q8 is not None:
                    return QT.wrap(a.shape, a.device, q8=a.q8, scale=a.scale * b)
                return QT.wrap(a.shape, a.device, node=Scaled(a, float(b)))
        if name == "cat" and ENABLED:
            tensors = args[0]
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            if all(isinstance(t, QT) and t.q8 is not None or not isinstance(t, QT) for t in tensors):
                shapes = [torch.empty(t.shape, device="meta") for t in tensors]
                meta = torch.cat(shapes, dim=dim)
                return QT.wrap(meta.shape, next(t.device for t in tensors if isinstance(t, QT)), node=Cat(list(tensors), dim))

        def real(t):
            return t.to_float() if isinstance(t, QT) else t

        args = torch.utils._pytree.tree_map(real, args)
        kwargs = torch.utils._pytree.tree_map(real, kwargs)
        return func(*args, **kwargs)


# ----------------------------------------------------------------------------------------------------------- pending nodes
class Node:
    def to_float(self):
        raise NotImplementedError


class Scaled(Node):
    """x * c on a pending node (attn * self.scale): only the scale changes, which the model code multiplies itself"""

    def __init__(self, x, c):
        self.x, self.c = x, c

    def to_float(self):
        return self.x.to_float() * self.c


class Cat(Node):
    def __init__(self, parts, dim):
        self.parts, self.dim = parts, dim

    def to_float(self):
        return torch.cat([p.to_float() if isinstance(p, QT) else p for p in self.parts], dim=self.dim)


class ModNode(Node):
    """the pending call of a module: `mod._slow(*float inputs)` reproduces what the ordinary path returns"""

    def __init__(self, kind, mod, shape, inputs, scales, out_scale):
        self.kind, self.mod, self.shape, self.inputs, self.scales, self.out_scale = kind, mod, tuple(shape), inputs, scales, out_scale

    def to_float(self):
        if self.kind == "linear":          # the classifier head: int32 GEMM and one conversion, nothing read back
            y = linear_to_float(self.mod, self.inputs[0], self.scales[0])
            if y is not None:
                return y
        xs = [x.to_float() if isinstance(x, QT) else x for x in self.inputs]
        return self.mod._slow(*xs, *self.scales)[0]


def pending(kind, mod, shape, device, inputs, scales, out_scale):
    """-> (QT holding the pending call, out_scale): what the module's forward returns"""
    return QT.wrap(shape, device, node=ModNode(kind, mod, shape, inputs, scales, out_scale)), out_scale


def q8_contig(x):
    """int8 payload of a QT as a contiguous tensor (None if x is not an int8-carrying QT)"""
    if isinstance(x, QT) and x.q8 is not None:
        return x.q8 if x.q8.is_contiguous() else x.q8.contiguous()
    return None


# ----------------------------------------------------------------------------------------------------------- the fused launches
def _cache(mod, key, build):
    c = mod.__dict__.setdefault("_lazy_cache", {})
    if key not in c:
        if len(c) >= 64:       # a model whose ranges keep changing (re-calibrated and re-frozen again and again): start over
            c.clear()
        c[key] = build()
    return c[key]


def _key(*hosts):
    return tuple(np.asarray(h, f32).tobytes() if h is not None else None for h in hosts)


def _frag_ok(N, K):
    """the weights-in-registers GEMM on v_mfma_i32_16x16x64_i8 (256- or 128-channel work items, chosen by the launcher)"""
    return K % 192 == 0 and N % 64 == 0 and N >= 128


def linear_consts(lin, s_in, device):
    """integer weights of a QuantLinear / QuantConv2d for input scale s_in (host float32): W8 row-major (+ the 16x16x64
    fragment copy where the weights-in-registers GEMM applies), b32, s_acc"""
    def build():
        lp = LinearParams(lin.weight.detach().cpu().numpy(), None if lin.bias is None else lin.bias.detach().cpu().numpy(), s_in)
        N, K = lp.W8.shape
        lin._publish(lp, device)           # the buffers the reference rewrites on every call
        d = dict(lp=lp, N=N, K=K, W=_dev(lp.W8, device), b=None if lp.b32 is None else _dev(lp.b32, device), Wf=None,
                 s_acc=QS.make(lp.s_acc, device))
        if K % 64 == 0 and _frag_ok(N, K):
            d["Wf"] = torch.empty((N + 63) // 64 * 64 * K, dtype=torch.int8, device=device)
            _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(d["W"]), K, N, K, _lib.ptr(d["Wf"]), _st())
        return d
    return _cache(lin, ("lin", lin.weight._version, None if lin.bias is None else lin.bias._version, _key(s_in), str(device)), build)


def _gemm_me(lin, s_in, s_out, device):
    """device (m, e) tables of the per-channel requantisation s_acc -> s_out, or None outside the kernels' contract (e < 31)"""
    c = linear_consts(lin, s_in, device)

    def build():
        m, e = dyadic(c["lp"].s_acc, s_out)
        if np.any(e < 31):
            return None
        return _dev(m.view(np.int32), device), _dev(e, device)
    # keyed on the weight / bias versions as linear_consts is: an in-place weight edit under unchanged ranges rebuilds W8 and
    # s_acc there, and the per-channel multipliers must follow (round-3 advisor finding: they did not)
    return _cache(lin, ("rq", lin.weight._version, None if lin.bias is None else lin.bias._version, _key(s_in, s_out), str(device)),
                  build)


def gemm_requant(lin, a8, s_in, s_out, device):
    """a8 [M, K] int8 contiguous -> int8 [M, N]: GEMM + per-channel requantisation to s_out in one kernel; None if outside the
    kernels' contract"""
    c = linear_consts(lin, s_in, device)
    N, K = c["N"], c["K"]
    if K % 64 != 0 or N % 16 != 0:
        return None
    me = _gemm_me(lin, s_in, s_out, device)
    if me is None:
        return None
    M = a8.shape[0]
    out = torch.empty(M, N, dtype=torch.int8, device=device)
    frags = c["Wf"] is not None and M >= 2048
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(a8), K, _lib.ptr(c["Wf"] if frags else c["W"]), K, _lib.ptr(c["b"]),
              _lib.ptr(me[0]), _lib.ptr(me[1]), _lib.ptr(out), N, M, N, K, 16 if frags else 0, _st())
    return out


def resolve(qact, x, pre_sf, identity, identity_sf, s_out, s_out_qs):
    """The frozen 8-bit QuantAct on a QT: one fused launch -> int8 QT.  None: not a recognised pattern (ordinary path)."""
    device = x.device
    s_in = host_of(pre_sf)
    if s_in is None:
        return None
    node = x.node
    out = None
    if isinstance(node, Requant) and x._q8 is None and identity is not None and not x.views:
        out = node.with_residual(identity, host_of(identity_sf), s_in, s_out, device)
        if out is not None:
            STATS["fused"] += 1
            return QT.wrap(out.shape, device, q8=out, scale=s_out_qs)
    if x.q8 is not None:
        if identity is None:
            return None
        x8, i8 = q8_contig(x), q8_contig(identity)
        s_id = host_of(identity_sf)
        if i8 is None or s_id is None or i8.shape != x8.shape or s_in.size != 1 or s_id.size != 1:
            return None
        (m1, e1), (m2, e2) = dyadic(s_in, s_out), dyadic(s_id, s_out)
        out = torch.empty_like(x8)
        _lib.call("ivit_residual_requant_i8", _lib.ptr(x8), int(m1[0]), int(e1[0]), _lib.ptr(i8), int(m2[0]), int(e2[0]),
                  _lib.ptr(out), x8.numel(), _st())
        STATS["fused"] += 1
        return QT.wrap(out.shape, device, q8=out, scale=s_out_qs)
    if isinstance(node, Cat):
        out = _resolve_cat(qact, node, s_in, identity, identity_sf, s_out, device)
    elif isinstance(node, ModNode) and identity is None:
        if s_in is not host_of(node.out_scale):     # the scale handed in must be the one the pending module returned
            return None
        if node.kind in ("linear", "conv"):
            s_a = host_of(node.scales[0])
            a8 = None
            if s_a is not None and s_a.size == 1:
                if node.kind == "conv":
                    a8 = _patchify_i8(node.mod, node.inputs[0])
                else:
                    a8 = q8_contig(node.inputs[0])
                    a8 = None if a8 is None else a8.reshape(-1, a8.shape[-1])
            if (a8 is not None and node.kind == "linear" and not x.views and a8.shape[0] >= 2048
                    and linear_consts(node.mod, s_a, device)["Wf"] is not None):
                # not launched yet: if the next QuantAct adds a residual (Block.qact2 / qact4, vit_quant.py:147,153) the GEMM,
                # this requantisation and that one are ONE kernel; any other consumer launches the GEMM as it is
                rq = Requant(node.mod, a8, s_a, s_out, node.shape, s_out_qs)
                if rq.ok:
                    return QT.wrap(x.shape, device, scale=s_out_qs, node=rq)
            o = gemm_requant(node.mod, a8, s_a, s_out, device) if a8 is not None else None
            if o is not None:
                sh = node.shape
                out = o.view(*sh) if node.kind == "linear" else o.view(sh[0], sh[2], sh[3], sh[1]).permute(0, 3, 1, 2)
        elif node.kind == "ln":
            out = _resolve_ln(node, s_out, device)
        elif node.kind == "ibln":
            out = _resolve_ibert_ln(node, s_out, device)
        elif node.kind == "gelu":
            out = _resolve_gelu(node, s_in, s_out, device)
        elif node.kind == "ibgelu":
            out = _resolve_ibert_gelu(node, s_in, s_out, device)
        elif node.kind == "matmul":
            out = _resolve_attention(node, s_in, s_out, device)
            if out is None:      # the first matmul of the attention chain: stays pending as the Shiftmax input
                base = node.inputs[0]
                if isinstance(base, QT) and not x.views:
                    return QT.wrap(x.shape, device, node=Scores(x, pre_sf, s_out, s_out_qs, qact))
    elif isinstance(node, Scaled) and identity is None and not x.views:
        return QT.wrap(x.shape, device, node=Scores(x, pre_sf, s_out, s_out_qs, qact))
    if out is None:
        return None
    STATS["fused"] += 1
    out = x.apply_views(out)
    return QT.wrap(out.shape, device, q8=out, scale=s_out_qs)


class Requant(Node):
    """a linear + its 8-bit QuantAct, not launched yet (see resolve)"""

    def __init__(self, lin, a8, s_a, s_out, shape, s_out_qs):
        self.lin, self.a8, self.s_a, self.s_out, self.shape, self.s_out_qs = lin, a8, s_a, s_out, shape, s_out_qs
        self.ok = _gemm_me(lin, s_a, s_out, a8.device) is not None
        self.out = None

    def force(self):
        if self.out is None:
            STATS["fused"] += 1
            self.out = gemm_requant(self.lin, self.a8, self.s_a, self.s_out, self.a8.device).view(*self.shape)
        return self.out

    def head_major(self, q, kT, v):
        """q, k^T and v of vit_quant.py:66-70 as recorded views of THIS linear's output [B, N, 3 H hd]: the GEMM writes them head-major
        ([3, B, H, N, hd], what the attention kernel reads) itself -> that tensor; None if the views are anything else"""
        if self.out is not None or len(self.shape) != 3 or any(t._q8 is not None or t.node is not self for t in (q, kT, v)):
            return None
        B, N, C3 = self.shape
        base = torch.empty(self.shape, dtype=torch.int8, device="meta")
        mq, mk, mv = q.apply_views(base), kT.apply_views(base).transpose(-2, -1), v.apply_views(base)
        if mq.dim() != 4:
            return None
        H, hd = mq.shape[1], mq.shape[3]
        C = H * hd
        if 3 * C != C3 or hd != 64 or N > 207 or any(tuple(m.shape) != (B, H, N, hd) or m.stride() != (N * C3, hd, C3, 1)
                                                     or m.storage_offset() != i * C for i, m in enumerate((mq, mk, mv))):
            return None
        device = self.a8.device
        c = linear_consts(self.lin, self.s_a, device)
        me = _gemm_me(self.lin, self.s_a, self.s_out, device)
        if c["Wf"] is None or c["K"] != C:
            return None
        hm = torch.empty(3, B, H, N, hd, dtype=torch.int8, device=device)
        _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(self.a8), c["K"], _lib.ptr(c["Wf"]), c["K"], _lib.ptr(c["b"]), _lib.ptr(me[0]),
                  _lib.ptr(me[1]), _lib.ptr(hm), N, H, hd, B * N, C3, c["K"], 16, _st())
        STATS["fused"] += 1
        return hm

    def to_float(self):
        return self.force().to(torch.float32) * self.s_out_qs.as_subclass(torch.Tensor).reshape(-1)[0]

    def with_residual(self, identity, s_id, s_in, s_out2, device):
        """out = clamp8(RNE(RNE(acc * M) * m1 / 2^e1) + RNE(identity * m2 / 2^e2)): ivit_gemm_i8_requant_residual_ex"""
        i8 = q8_contig(identity)
        c = linear_consts(self.lin, self.s_a, device)
        N, K = c["N"], c["K"]
        if (i8 is None or s_id is None or s_id.size != 1 or s_in.size != 1 or s_in[0] != f32(self.s_out) or c["Wf"] is None
                or tuple(i8.shape) != tuple(self.shape)):
            return None
        me = _gemm_me(self.lin, self.s_a, self.s_out, device)
        (m1, e1), (m2, e2) = dyadic(s_in, s_out2), dyadic(s_id, s_out2)
        M = self.a8.shape[0]
        out = torch.empty(M, N, dtype=torch.int8, device=device)
        _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(self.a8), K, _lib.ptr(c["Wf"]), K, _lib.ptr(c["b"]), _lib.ptr(me[0]),
                  _lib.ptr(me[1]), _lib.ptr(i8), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(out), N, M, N, K, 16, _st())
        return out.view(*self.shape)


class Scores(Node):
    """qact_attn1 on the (scaled) q . k^T: pending until Shiftmax and the second matmul arrive (vit_quant.py:72-82)"""

    def __init__(self, x, pre_sf, s_out, s_out_qs, qact):
        self.x, self.pre_sf, self.s_in, self.s_out, self.s_out_qs, self.qact = x, pre_sf, pre_sf.host, s_out, s_out_qs, qact

    def to_float(self):
        return self.qact._slow(self.x.to_float(), self.pre_sf)[0]


class Probs(Node):
    def __init__(self, scores_qt, mod):
        self.x, self.mod = scores_qt, mod

    def to_float(self):
        sc = self.x.node
        return self.mod._slow(self.x.to_float(), sc.s_out_qs)[0]


def _patchify_i8(conv, x):
    """[B, Cin, H, W] int8 QT -> the im2col operand [B * g * g, Cin * k * k] of the non-overlapping patch convolution"""
    x8 = x.q8 if isinstance(x, QT) else None
    if x8 is None:
        return None
    kh, kw = conv.kernel_size
    B, Cin, H, W = x8.shape
    if not (kh == kw == conv.stride[0] == conv.stride[1] and conv.padding == (0, 0) and H % kh == 0 and W % kw == 0):
        return None
    g, h = H // kh, W // kw
    return x8.reshape(B, Cin, g, kh, h, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * g * h, Cin * kh * kw)


def _resolve_ln(node, s_out, device):
    ln, x = node.mod, node.inputs[0]
    x8 = q8_contig(x)
    s_in = host_of(node.scales[0])
    if x8 is None or s_in is None or s_in.size != 1:
        return None
    C = x8.shape[-1]

    def build():
        lp = LayerNormParams(ln.weight.detach().cpu().numpy(), ln.bias.detach().cpu().numpy(), s_out)
        return dict(bias=_dev(lp.bias_int, device), s=_dev(lp.s_ln, device), m=_dev(lp.m.view(np.int32), device), e=_dev(lp.e, device))
    try:
        c = _cache(ln, ("ln", ln.weight._version, ln.bias._version, _key(s_out), str(device)), build)
    except ValueError:
        return None
    tabs = _cache(ln, ("phi", _key(s_in), str(device)),
                  lambda: (lambda t: None if t is None else (_dev(t[0], device), _dev(t[1], device)))(phi_tables(s_in[0])))
    rows = x8.numel() // C
    out = torch.empty_like(x8)
    if tabs is not None:
        _lib.call("ivit_layernorm_i8_compat", _lib.ptr(x8), C, rows, C, _lib.ptr(c["bias"]), _lib.ptr(c["s"]), _lib.ptr(c["m"]),
                  _lib.ptr(c["e"]), _lib.ptr(tabs[0]), _lib.ptr(tabs[1]), _lib.ptr(out), C, 0, _st())
    else:
        _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x8), C, rows, C, _lib.ptr(c["bias"]), _lib.ptr(c["s"]), _lib.ptr(c["m"]),
                  _lib.ptr(c["e"]), _lib.ptr(out), C, 0, _st())
    return out


def _resolve_ibert_ln(node, s_out, device):
    """IBERTIntLayerNorm (ibert_modules.py:126-153) + the QuantAct behind it on int8: ivit_ibert_layernorm_i8 (csrc/ibert.hip), which
    works on fl(q * s_in) literally -- any input scale"""
    ln, x = node.mod, node.inputs[0]
    x8 = q8_contig(x)
    s_in = host_of(node.scales[0])
    if x8 is None or s_in is None or s_in.size != 1 or ln.overflow_handling:
        return None
    C = x8.shape[-1]

    def build():
        lp = LayerNormParams(ln.weight.detach().cpu().numpy(), ln.bias.detach().cpu().numpy(), s_out)
        return dict(bias=_dev(lp.bias_int, device), s=_dev(lp.s_ln, device), m=_dev(lp.m.view(np.int32), device), e=_dev(lp.e, device),
                    shift_pow2=float(2.0 ** float(ln.shift.reshape(-1)[0])))
    try:
        c = _cache(ln, ("ibln", ln.weight._version, ln.bias._version, id(ln.shift), ln.shift._version, _key(s_out), str(device)), build)
    except ValueError:
        return None
    out = torch.empty_like(x8)
    _lib.call("ivit_ibert_layernorm_i8", _lib.ptr(x8), C, x8.numel() // C, C, float(s_in[0]), _lib.ptr(c["bias"]), _lib.ptr(c["s"]),
              c["shift_pow2"], _lib.ptr(c["m"]), _lib.ptr(c["e"]), _lib.ptr(out), C, 0, _st())
    return out


def _resolve_ibert_gelu(node, s_g_out, s_out, device):
    """IBERTIntGELU (ibert_modules.py:205-232) + mlp.qact1: a map of the input byte alone, as a table (ivit_ibert_gelu_build_lut;
    replicated over the row-maximum axis of the ShiftGELU table format).  The GELU's output scale is negative: requant(z, s) ==
    requant(-z, -s), handled inside the table build"""
    x8 = q8_contig(node.inputs[0])
    s_g = host_of(node.scales[0])
    if x8 is None or s_g is None or s_g.size != 1:
        return None

    def build():
        from .ibert_modules import gelu_constants
        gb, gc, gsh, gso = gelu_constants(s_g[0])
        mg, eg = dyadic(abs(f32(gso)), s_out)
        lut = torch.empty(65536, dtype=torch.int8, device=device)
        _lib.call("ivit_ibert_gelu_build_lut", float(s_g[0]), gb, gc, gsh, float(gso), int(mg[0]), int(eg[0]), _lib.ptr(lut), _st())
        return lut
    lut = _cache(node.mod, ("ibgelu", _key(s_g, s_out), str(device)), build)
    L = x8.shape[-1]
    out = torch.empty_like(x8)
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(x8), L, x8.numel() // L, L, _lib.ptr(lut), _lib.ptr(out), L, 0, _st())
    return out


def _resolve_gelu(node, s_g_out, s_out, device):
    x8 = q8_contig(node.inputs[0])
    s_g = host_of(node.scales[0])
    if x8 is None or s_g is None or s_g.size != 1:
        return None

    def build():
        mg, eg = dyadic(f32(s_g[0] * f32(1.0 / 128.0)), s_out)              # ivit_modules.py:121,124
        lut = torch.empty(65536, dtype=torch.int8, device=device)
        t = phi_tables(s_g[0])
        remap = None if t is None else _dev(t[0], device)
        _lib.call("ivit_shiftgelu_build_lut_ex", float(s_g[0]), int(mg[0]), int(eg[0]), _lib.ptr(remap), _lib.ptr(lut), _st())
        return lut
    lut = _cache(node.mod, ("gelu", _key(s_g, s_out), str(device)), build)
    L = x8.shape[-1]
    out = torch.empty_like(x8)
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(x8), L, x8.numel() // L, L, _lib.ptr(lut), _lib.ptr(out), L, 0, _st())
    return out


def _resolve_attention(node, s_pv, s_out, device):
    """matmul_2(probs, v) behind qact2, where probs = Shiftmax(qact_attn1(matmul_1(q, k^T) * scale)): the fused attention kernel"""
    P, v = node.inputs
    if not (isinstance(P, QT) and isinstance(P.node, Probs) and not P.views and isinstance(v, QT)):
        return None
    sc_qt = P.node.x
    sc = sc_qt.node
    scaled = sc.x.node if isinstance(sc.x.node, Scaled) else None
    mm = scaled.x.node if scaled is not None else sc.x.node
    if not (isinstance(mm, ModNode) and mm.kind == "matmul") or (scaled is not None and scaled.x.views):
        return None
    s_mm = host_of(mm.out_scale)          # the scale the scores arrive with: s_q * s_k, times the model's factor
    if s_mm is None or not np.array_equal(sc.s_in, s_mm if scaled is None else (s_mm * f32(scaled.c)).astype(f32)):
        return None
    q, kT = mm.inputs
    if not (isinstance(q, QT) and isinstance(kT, QT)):
        return None
    hm = q.node.head_major(q, kT, v) if isinstance(q.node, Requant) else None
    if hm is not None:
        B, H, T, hd = hm.shape[1:]
    else:
        if q.q8 is None or kT.q8 is None or v.q8 is None:
            return None
        q8, k8, v8 = q.q8, kT.q8.transpose(-2, -1), v.q8
        if q8.dim() != 4 or q8.shape != k8.shape or q8.shape != v8.shape or q8.shape[-1] != 64 or q8.shape[-2] > 207:
            return None
        B, H, T, hd = q8.shape
        hm = torch.empty(3, B, H, T, hd, dtype=torch.int8, device=device)
        hm[0].copy_(q8)
        hm[1].copy_(k8)
        hm[2].copy_(v8)
    s_S, s_at = sc.s_in, sc.s_out
    if s_S.size != 1 or s_pv.size != 1:
        return None
    sm = P.node.mod
    if type(sm).__name__ == "IBERTIntSoftmax":
        # IBERTIntSoftmax (ibert_modules.py:237-319): exp_int after its internal 16-bit QuantAct as a (row max, q) table, row sum in
        # torch's float32 order inside the kernel (attention.hip MODE 3 / 4); the kernel holds 193 .. 207 tokens
        if not (192 < T < 208) or sm.output_bit != 8 or sm.act.running_stat:
            return None

        def build_ib():
            from .ibert_modules import softmax_constants
            lo, hi = float(sm.act.x_min.reshape(-1)[0]), float(sm.act.x_max.reshape(-1)[0])
            x0i, bi, ci, exp_sf, act_sf, ma, ea = softmax_constants(s_at, lo, hi)
            tab = torch.empty(65536, dtype=torch.float32, device=device)
            _lib.call("ivit_ibert_softmax_build_table", float(s_at), x0i, bi, ci, float(exp_sf), float(act_sf), ma, ea, _lib.ptr(tab), _st())
            ms, mo = dyadic(s_S, s_at), dyadic(s_pv, s_out)
            d = dict(ms=(int(ms[0][0]), int(ms[1][0])), mo=(int(mo[0][0]), int(mo[1][0])), tab=tab, band=None, band_w=0)
            band, bw = shiftexp_band(tab.cpu().numpy().view(np.uint32).reshape(256, 256))
            if bw and bw <= 128:
                d.update(band=_dev(band.view(np.float32), device), band_w=bw)
            sm.act.act_scaling_factor = torch.full((1,), float(act_sf), dtype=torch.float32, device=device)
            return d
        a = _cache(sc.qact, ("ibattn", _key(s_S, np.asarray(s_at), s_pv, np.asarray(s_out)), id(sm.act.x_min), sm.act.x_min._version,
                             id(sm.act.x_max), sm.act.x_max._version, str(device)), build_ib)
        out = torch.empty(B * T, H * hd, dtype=torch.int8, device=device)
        _lib.call("ivit_attention_fused_i8_ibert", _lib.ptr(hm), _lib.ptr(out), B, H, T, hd, a["ms"][0], a["ms"][1], a["mo"][0], a["mo"][1],
                  _lib.ptr(a["tab"]), _lib.ptr(a["band"]), a["band_w"], 0, _st())
        return out.view(B, T, H, hd).permute(0, 2, 1, 3)

    def build():
        ms, mo = dyadic(s_S, s_at), dyadic(s_pv, s_out)
        d = dict(ms=(int(ms[0][0]), int(ms[1][0])), mo=(int(mo[0][0]), int(mo[1][0])), exp2d=None, band=None, band_w=0)
        if phi_tables(s_at) is not None:
            tab = shiftexp2d(s_at)
            band, bw = shiftexp_band(tab)
            if bw and bw <= 128:
                d.update(band=_dev(band.view(np.int32), device), band_w=bw)
            else:
                d["exp2d"] = _dev(tab.view(np.int32), device)
        return d
    a = _cache(sc.qact, ("attn", _key(s_S, np.asarray(s_at), s_pv, np.asarray(s_out)), str(device)), build)
    out = torch.empty(B * T, H * hd, dtype=torch.int8, device=device)
    _lib.call("ivit_attention_fused_i8_compat_band", _lib.ptr(hm), _lib.ptr(out), B, H, T, hd, a["ms"][0], a["ms"][1], float(s_at),
              a["mo"][0], a["mo"][1], _lib.ptr(a["exp2d"]), _lib.ptr(a["band"]), a["band_w"], 0, _st())
    return out.view(B, T, H, hd).permute(0, 2, 1, 3)


def _resolve_cat(qact, node, s_in, identity, identity_sf, s_out, device):
    """qact1(cat(cls_token, patches), s, pos, s_pos) (vit_quant.py:293-297): the raw float rows go through round(x / s), the
    int8 part is widened, then the two-operand requantisation"""
    s_id = host_of(identity_sf)
    if s_in.size != 1 or node.dim != 1 or (identity is not None and (s_id is None or s_id.size != 1)):
        return None
    parts = []
    for p in node.parts:
        if isinstance(p, QT):
            parts.append(p.q8.to(torch.int32))
        else:
            pf = p.detach()
            z = torch.round(pf.to(torch.float32) / float(s_in[0])).to(torch.int32)      # quant_utils.py:220
            parts.append(z)
    z = torch.cat(parts, dim=1).contiguous()
    C = z.shape[-1]
    m, e = dyadic(s_in, s_out)
    md, ed = _cache(qact, ("me", _key(s_in, np.asarray(s_out)), str(device)), lambda: (_dev(m.view(np.int32), device), _dev(e, device)))
    z2 = m2d = e2d = None
    n2 = 0
    if identity is not None:
        i8 = identity.q8 if isinstance(identity, QT) else None
        if i8 is None:
            return None
        z2 = i8.to(torch.int32).expand_as(z).contiguous()
        m2, e2 = dyadic(s_id, s_out)
        m2d, e2d = _cache(qact, ("me2", _key(s_id, np.asarray(s_out)), str(device)), lambda: (_dev(m2.view(np.int32), device), _dev(e2, device)))
        n2 = 1
    q = torch.empty_like(z)
    _lib.call("ivit_requant_i32", _lib.ptr(z), z.numel() // C, C, _lib.ptr(md), _lib.ptr(ed), 1, _lib.ptr(z2), _lib.ptr(m2d),
              _lib.ptr(e2d), n2, 8, _lib.ptr(q), _st())
    return q.to(torch.int8)


def linear_to_float(lin, x, s_in_qs):
    """a pending linear read as floats (the classifier head): int32 GEMM, then acc * s_acc"""
    a8 = q8_contig(x)
    s_in = host_of(s_in_qs)
    if a8 is None or s_in is None or s_in.size != 1:
        return None
    c = linear_consts(lin, s_in, a8.device)
    N, K = c["N"], c["K"]
    if K % 64 != 0:
        return None
    Np = (N + 3) // 4 * 4
    if Np != N:
        return None
    a2 = a8.reshape(-1, K)
    acc = torch.empty(a2.shape[0], N, dtype=torch.int32, device=a8.device)
    _lib.call("ivit_gemm_i8_i32", _lib.ptr(a2), K, _lib.ptr(c["W"]), K, _lib.ptr(c["b"]), _lib.ptr(acc), N, a2.shape[0], N, K, _st())
    y = torch.empty(acc.shape, dtype=torch.float32, device=a8.device)
    s = c["s_acc"].as_subclass(torch.Tensor)
    _lib.call("ivit_i32_to_f32", _lib.ptr(acc), acc.shape[0], N, _lib.ptr(s), N, _lib.ptr(y), _st())
    return y.view(*x.shape[:-1], N)
