"""Integer-only Swin forward on MI355X (config 5): int8 GEMM operands, int16 residual stream, every operator a
hand-written HIP kernel reached through the C ABI (include/ivit_hip.h, second half).

Dataflow = the reference's frozen-model forward (/root/reference/models/swin_quant.py:539-564; WindowAttention
:121-169; SwinTransformerBlock :251-301; PatchMerging :328-349; PatchEmbed with norm, layers_quant.py:191-203):

  images f32 --quantize+im2col(4x4)--> int8 [B*3136, 48|64] --GEMM+requant--> qact_before_norm int8
    --I-LayerNorm+requant--> patch_embed.qact int8 --requant 8->16--> x int16 [B*3136, 96]
  per block { LN16->8 written in window order (partition + cyclic shift are a row map inside the kernel)
              -> GEMM qkv (+requant, head-major per window) -> window attention (bias, mask, Shiftmax, P.V)
              -> GEMM proj (int32) -> qact4(16) + window reverse + residual QuantAct(16) in one pass
              -> LN16->8 -> GEMM fc1 -> ShiftGELU table -> GEMM fc2 (+requant) -> residual QuantAct(16) }
  per stage end { 2x2 patch-merge gather int16 -> LN16->8 -> GEMM reduction (+requant) -> widen to int16 }
  LN16->8 -> token average pool + requant -> GEMM head -> INT32 logits -> scale + argmax

There is no fallback path: a missing libivit_hip.so or a kernel error raises.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .graph import GraphReplay
from .prepare import (IMAGENET_MEAN, IMAGENET_STD, LayerNormParams, LinearParams, dyadic, f32, input_lut_u8, markstein_division_ok, pad_head, phi_is_identity, phi_table, window_shiftexp_band,
                      phi_tables, quant_sym,
                      requant_host, sym_scale)
from .synth import IMG_SIZE

PATCH = 4
HEAD_DIM = 32
IDENT = (1 << 30, 30)  # dyadic 1.0


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def _pad64(k):
    return (k + 63) // 64 * 64


def rel_position_index(ws: int) -> np.ndarray:
    """relative_position_index of WindowAttention (swin_quant.py:73-88): [ws*ws, ws*ws] into the (2ws-1)^2 table."""
    yy, xx = np.divmod(np.arange(ws * ws), ws)
    dy = yy[:, None] - yy[None, :] + (ws - 1)
    dx = xx[:, None] - xx[None, :] + (ws - 1)
    return dy * (2 * ws - 1) + dx


def shift_mask_regions(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """Region id of every window token of a shifted block (swin_quant.py:223-243): [nW, ws*ws]."""
    def band(n):
        b = np.zeros(n, np.int64)
        b[n - ws:n - shift] = 1
        b[n - shift:] = 2
        return b
    img = band(H)[:, None] * 3 + band(W)[None, :]
    return img.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)


def window_row_map(B: int, H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """Host restatement of the kernels' win_row (csrc/swin.hip): destination row, in window order, of token (b,y,x)."""
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ys, xs = (y - shift) % H, (x - shift) % W
    idx = ((ys // ws) * (W // ws) + xs // ws) * (ws * ws) + (ys % ws) * ws + xs % ws
    return (np.arange(B)[:, None] * (H * W) + idx.reshape(-1)[None, :]).reshape(-1)


class IntSwinEngine(GraphReplay):
    def __init__(self, float_state, ranges, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=7,
                 device="cuda:0", max_batch: int = 64):
        self.C0, self.depths, self.heads, self.window = embed_dim, tuple(depths), tuple(num_heads), window
        self.dev = torch.device(device)
        self.max_batch = max_batch
        _lib.lib()  # fail loudly now if the HIP library is absent
        P = {k: _np(v).astype(np.float32) for k, v in float_state.items()}
        R = ranges

        def s(name, bits=8):
            lo, hi = R[name]
            return sym_scale(lo, hi, bits)

        def dev(a, dtype=None):
            t = torch.from_numpy(np.ascontiguousarray(a))
            if dtype is not None:
                t = t.to(dtype)
            return t.to(self.dev)

        def lin_host(name, s_in):
            lp = LinearParams(P[name + ".weight"], P.get(name + ".bias"), s_in)
            Kp = _pad64(lp.K)
            W = np.zeros((lp.W8.shape[0], Kp), np.int8)   # zero K-padding: the operand's pad columns never contribute
            W[:, :lp.K] = lp.W8
            d = dict(W=dev(W), b=None if lp.b32 is None else dev(lp.b32), K=Kp, N=W.shape[0], Wb=None)
            if d["N"] >= 128 and d["N"] % 16 == 0:
                # block-layout copy (include/ivit_hip.h IVIT_LAYOUT_BLOCKS) for the calls that reach the persistent GEMM
                d["Wb"] = torch.empty_like(d["W"])
                _lib.call("ivit_tile_operand_i8", _lib.ptr(d["W"]), Kp, d["N"], Kp, _lib.ptr(d["Wb"]), _lib.stream_ptr())
            d["Wf"] = None
            N = d["N"]
            wf_bit = 8 if name.endswith("attn.proj") else 16
            narrow_ok = name.endswith(("attn.qkv", "mlp.fc1"))      # int8 epilogues; mlp.fc2 (16-bit residual epilogue) has full tiles only
            if Kp % 192 == 0 and N % 64 == 0 and N >= 128 and (narrow_ok or (N + 255) // 256 * 256 * 8 <= N * 9):
                # MFMA-fragment copy: the weights-in-registers GEMM (qkv / fc1 of stages 1-3, all of stage 3).  The 16x16x64
                # order (IVIT_W_FRAGS16; 128-channel work items where 256-channel tiles fit badly, round 4) except for attn.proj,
                # whose fused 16-bit epilogue exists for the 32x32x32 form only
                d["Wf_bit"] = wf_bit
                d["Wf"] = torch.empty((N + 63) // 64 * 64 * Kp, dtype=torch.int8, device=self.dev)
                _lib.call("ivit_pack_weight_frags_i8" if d["Wf_bit"] == 8 else "ivit_pack_weight_frags16_i8", _lib.ptr(d["W"]), Kp, N, Kp,
                          _lib.ptr(d["Wf"]), _lib.stream_ptr())
            return lp, d

        def lin_dev(name, s_in, s_out):
            lp, d = lin_host(name, s_in)
            m, e = lp.requant_to(s_out)
            d.update(m=dev(m.view(np.int32)), e=dev(e))
            return d

        self.proj_fused = True     # attention output in image order + attn.proj / attn.qact4 / qact2 in one GEMM (False: A/B, tests)
        self.proj_i16 = True       # attn.proj writes the 16-bit attn.qact4 output instead of raw accumulators
        self.natural_sites = 0     # operators whose input scale is not a power of two: literal / table-driven kernels
        self.window_softmax_forms = []     # per natural-scale attention block: "band1xW" / "band256xW" / "literal" (prepare.window_shiftexp_band)

        def ln_dev(prefix, s_out, s_in, bits_in=16):
            """s_in: scale of the LayerNorm's input.  If fl(fl(q*s_in)/s_in) != q for some q of that width, the reference's
            LayerNorm sees those neighbouring floats (ivit_modules.py:36-38): 16-bit inputs take the literal kernel, the 8-bit
            patch norm the table form of the DeiT engine."""
            lp = LayerNormParams(P[prefix + ".weight"], P[prefix + ".bias"], s_out)
            d = dict(bias=dev(lp.bias_int), s=dev(lp.s_ln), m=dev(lp.m.view(np.int32)), e=dev(lp.e), s_in=None, remap=None, phi=None,
                     fast_div=0)
            if not phi_is_identity(s_in, bits_in):
                self.natural_sites += 1
                d["s_in"] = float(s_in)
                d["fast_div"] = int(bits_in == 16 and markstein_division_ok(s_in, 16))
                if bits_in == 8:
                    remap, phi = phi_tables(s_in)
                    d.update(remap=dev(remap), phi=dev(phi))
            return d

        def sme(pre, z):
            m, e = dyadic(pre, z)
            return int(m[0]), int(e[0])

        # ---- stem (layers_quant.py:191-203 with norm_layer; swin_quant.py:541-546)
        s0 = s("qact_input")
        self.inv_s0 = float(f32(1.0) / s0)
        self.s0 = float(s0)
        self.input_lut = None
        s_bn = s("patch_embed.qact_before_norm")
        self.patch = lin_dev("patch_embed.proj", s0, s_bn)
        s_pq = s("patch_embed.qact")
        self.patch_ln = ln_dev("patch_embed.norm", s_pq, s_bn, 8)
        s_x = s("qact1", 16)
        self.stem_me = sme(s_pq, s_x)

        # ---- stages
        self.stages = []
        G = IMG_SIZE // PATCH
        H = W = G
        C = embed_dim
        for li, (depth, nH) in enumerate(zip(self.depths, self.heads)):
            if C // nH != HEAD_DIM:
                raise ValueError("window attention kernel supports head_dim 32 only")
            st = dict(H=H, W=W, C=C, nH=nH, blocks=[], down=None)
            for bi in range(depth):
                p = f"layers.{li}.blocks.{bi}."
                win = min(window, H)
                shift = 0 if (bi % 2 == 0 or min(H, W) <= window) else window // 2
                N = win * win
                blk = dict(win=win, shift=shift)
                s_q1 = s(p + "qact1")
                blk["ln1"] = ln_dev(p + "norm1", s_q1, s_x)
                s_a1 = s(p + "attn.qact1")
                blk["qkv"] = lin_dev(p + "attn.qkv", s_q1, s_a1)
                s_S = f32(f32(s_a1 * s_a1) * f32(HEAD_DIM ** -0.5))            # swin_quant.py:139-141
                s_at = s(p + "attn.qact_attn1")
                s_tab = s(p + "attn.qact_table")
                s_A = s(p + "attn.qact2")
                ktab = quant_sym(P[p + "attn.relative_position_bias_table"], s_tab, 8)   # [(2ws-1)^2, nH]
                m2, e2 = dyadic(s_tab, s_A)
                bias = ktab[rel_position_index(win).reshape(-1)].reshape(N, N, nH).transpose(2, 0, 1)
                bias_add = requant_host(bias, m2[0], e2[0])                    # identity operand of qact2, :143-147
                assert np.abs(bias_add).max() < 32768
                bias_pad = np.zeros((nH, N, 64), np.int16)
                bias_pad[:, :, :N] = bias_add
                region, mask_value = None, 0
                # Shiftmax input: phi(q) = fl(fl(q*s)/s) for a plain score, fl(fl(fl(q*s) - 100)/s) for one under the shift
                # mask (:149-156 adds float -100 to q*s, ivit_modules.py:165 divides by s).  Integer kernel when phi is the
                # identity and -100/s an integer; else the literal float sequence on the two 256-entry tables
                qv = np.arange(-128, 128, dtype=f32)
                phi_m = ((((qv * s_A).astype(f32) + f32(-100.0)).astype(f32)) / s_A).astype(f32)
                att_nat = not phi_is_identity(s_A)
                if shift:
                    mval = f32(-100.0) / s_A                                    # :149-155: (k*s + (-100)) / s
                    if mval != np.rint(mval) or abs(mval) >= 32768:
                        att_nat = True
                        mask_value = -1                                         # unused by the literal form
                    else:
                        mask_value = int(mval)
                    region = np.zeros(((H // win) * (W // win), 64), np.uint8)
                    region[:, :N] = shift_mask_regions(H, W, win, shift)
                band, band_w = (None, 0)
                if att_nat:
                    self.natural_sites += 1
                    # table form of the natural-scale Shiftmax where it is provably what the reference computes (every masked score
                    # saturated, no masked row maximum); else the kernel's literal float sequence on phi / phi_m
                    band, band_w = window_shiftexp_band(s_A, bool(shift))
                    self.window_softmax_forms.append("literal" if band is None else f"band{band.shape[0]}x{band_w}")
                s_pv = f32(f32(1.0 / 128.0) * s_a1)
                s_a3 = s(p + "attn.qact3")
                blk["attn"] = dict(ms=sme(s_S, s_at), mb=sme(s_at, s_A), s_attn=float(s_A), mo=sme(s_pv, s_a3),
                                   bias=dev(bias_pad), region=None if region is None else dev(region),
                                   mask_value=mask_value, nW=(H // win) * (W // win),
                                   phi=dev(phi_table(s_A)) if att_nat else None, phim=dev(phi_m) if att_nat else None,
                                   band=None if band is None else dev(band), band_w=band_w)
                lp, d = lin_host(p + "attn.proj", s_a3)
                s_a4 = s(p + "attn.qact4", 16)
                mp, ep = dyadic(lp.s_acc, s_a4)
                d.update(m=dev(mp.view(np.int32)), e=dev(ep))
                blk["proj"] = d
                s_b2 = s(p + "qact2", 16)
                blk["res1"] = sme(s_a4, s_b2) + sme(s_x, s_b2)
                s_b3 = s(p + "qact3")
                blk["ln2"] = ln_dev(p + "norm2", s_b3, s_b2)
                s_g = s(p + "mlp.qact_gelu")
                blk["fc1"] = lin_dev(p + "mlp.fc1", s_b3, s_g)
                s_go = f32(s_g * f32(1.0 / 128.0))
                s_m1 = s(p + "mlp.qact1")
                mg, eg = sme(s_go, s_m1)
                lut = torch.empty(65536, dtype=torch.int8, device=self.dev)
                g_tabs = phi_tables(s_g)            # ShiftGELU sees trunc(phi(q)) (ivit_modules.py:106-107)
                g_remap = None
                if g_tabs is not None:
                    self.natural_sites += 1
                    g_remap = dev(g_tabs[0])
                _lib.call("ivit_shiftgelu_build_lut_ex", float(s_g), mg, eg, _lib.ptr(g_remap), _lib.ptr(lut), self._stream())
                blk["gelu_lut"] = lut
                s_m2 = s(p + "mlp.qact2")
                blk["fc2"] = lin_dev(p + "mlp.fc2", s_m1, s_m2)
                s_b4 = s(p + "qact4", 16)
                blk["res2"] = sme(s_m2, s_b4) + sme(s_b2, s_b4)
                s_x = s_b4
                st["blocks"].append(blk)
            if li < len(self.depths) - 1:
                p = f"layers.{li}.downsample."
                s_d1 = s(p + "qact1")
                s_d2 = s(p + "qact2")
                st["down"] = dict(ln=ln_dev(p + "norm", s_d1, s_x), red=lin_dev(p + "reduction", s_d1, s_d2))
                s_x = s_d2
            self.stages.append(st)
            if st["down"] is not None:
                H, W, C = H // 2, W // 2, 2 * C
        self.C_last, self.T_last = C, H * W

        # ---- tail (swin_quant.py:552-563)
        s_q2 = s("qact2")
        self.ln_f = ln_dev("norm", s_q2, s_x)
        s_q3 = s("qact3")
        self.pool_me = sme(s_q2, s_q3)
        lp = LinearParams(P["head.weight"], P.get("head.bias"), s_q3)
        hW, hb, hs, self.num_classes = pad_head(lp.W8, lp.b32, lp.s_acc)      # any class count
        self.head = dict(W=dev(hW), b=dev(hb), K=lp.K, N=hW.shape[0])
        self.head_scale = dev(hs)
        self._alloc(max_batch)
        self._compact(True)
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return _lib.stream_ptr()

    def _alloc(self, B):
        C0 = self.C0
        M0 = B * (IMG_SIZE // PATCH) ** 2
        ld0 = _pad64(C0)
        i8 = dict(dtype=torch.int8, device=self.dev)
        i16 = dict(dtype=torch.int16, device=self.dev)
        slack = 256  # GEMM K-padding reads up to ld - C bytes past the last row of a C-strided operand
        self.ws = dict(
            a0=torch.zeros(M0 * 64 + slack, **i8),
            pe=torch.empty(M0 * C0, **i8), pn=torch.empty(M0 * C0, **i8),
            x=torch.empty(M0 * C0, **i16), x2=torch.empty(M0 * C0, **i16),
            h=torch.zeros(M0 * ld0 + slack, **i8), ao=torch.zeros(M0 * ld0 + slack, **i8),
            qkv=torch.empty(3 * M0 * C0, **i8),
            acc=torch.empty(M0 * C0, dtype=torch.int32, device=self.dev),
            f1=torch.empty(M0 * 4 * C0, **i8), g=torch.empty(M0 * 4 * C0, **i8), f2=torch.empty(M0 * C0, **i8),
            xm=torch.empty(M0 * C0, **i16), hm=torch.empty(M0 * C0, **i8), red=torch.empty(M0 * C0 // 2, **i8),
            hN=torch.empty(B * self.T_last * self.C_last, **i8), pooled=torch.empty(B * self.C_last, **i8),
            logits=torch.empty(B, self.head["N"], dtype=torch.int32, device=self.dev),
            logits_f=torch.empty(B, self.head["N"], dtype=torch.float32, device=self.dev),
            top1=torch.empty(B, dtype=torch.int32, device=self.dev),
        )

    def _compact(self, on=True):
        """Alias workspaces whose lifetimes do not overlap (see IntViTEngine._compact): the 16-bit residual QuantActs and the
        GELU run in place, the attention output reuses the LayerNorm buffer, q/k/v and the projection's int32 accumulators
        live inside the fc1 / GELU buffer.  One stage-0 block touches ~330 MB instead of ~870 MB at batch 128."""
        ws = self.ws
        if "_own" not in ws:
            ws["_own"] = {k: ws[k] for k in ("x2", "ao", "qkv", "acc", "g")}
        if on:
            ws["x2"] = ws["x"]
            ws["ao"] = ws["h"]
            ws["g"] = ws["f1"]
            ws["qkv"] = ws["f1"][: ws["_own"]["qkv"].numel()]
            ws["acc"] = ws["f1"].view(torch.int32)[: ws["_own"]["acc"].numel()]
        else:
            ws.update(ws["_own"])

    @staticmethod
    def _w(lin, M):
        """(weight pointer, layouts) -- the block-layout copy when the call goes to the persistent kernel"""
        if lin.get("Wf") is not None and M >= 2048:
            return _lib.ptr(lin["Wf"]), lin["Wf_bit"]
        if lin["K"] <= 128 and lin["N"] <= 320 and M >= 8192:
            return _lib.ptr(lin["W"]), 0          # row-major operands: the skinny-K form (csrc/gemm.hip, round 4) keeps the whole W in LDS
        if lin["Wb"] is not None and M >= 2048:
            return _lib.ptr(lin["Wb"]), 2
        return _lib.ptr(lin["W"]), 0

    def _gemm(self, A, lda, lin, out, ldo, M, st):
        w, lay = self._w(lin, M)
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(A), lda, w, lin["K"], _lib.ptr(lin["b"]),
                  _lib.ptr(lin["m"]), _lib.ptr(lin["e"]), _lib.ptr(out), ldo, M, lin["N"], lin["K"], lay, st)

    def _ln16(self, x, rows, C, ln, out, ldo, st, H=0, W=0, ws=0, shift=0, outer=0):
        if ln["s_in"] is not None:       # natural input scale: the literal kernel (csrc/swin.hip)
            _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(x), rows, C, ln["s_in"], ln["fast_div"] | (outer << 8), _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), ldo, H, W, ws, shift, st)
            return
        _lib.call("ivit_layernorm_i16_i8", _lib.ptr(x), rows, C, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                  _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), ldo, H, W, ws, shift, st)

    # ------------------------------------------------------------------ forward
    def set_input_normalisation(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        """uint8 input (see IntViTEngine.set_input_normalisation): the Normalize transform folded into the input table"""
        self.input_lut = torch.from_numpy(input_lut_u8(self.s0, mean, std)).to(self.dev)

    def forward(self, images: torch.Tensor, taps: dict | None = None):
        """images: float32 [B,3,224,224] on the engine's device.  Returns (logits_int32 [B,1000], logits_f32, top1)
        -- views of the engine's workspace, valid until the next call.  `taps` (tests) receives clones of the
        intermediate integer tensors in the reference's layouts."""
        assert images.is_cuda and images.dtype in (torch.float32, torch.uint8) and images.is_contiguous()
        B = images.shape[0]
        assert images.shape[1:] == (3, IMG_SIZE, IMG_SIZE) and 0 < B <= self.max_batch
        ws = self.ws
        st = self._stream()
        G = IMG_SIZE // PATCH
        C0 = self.C0
        M = B * G * G

        def tap(name, t, rows, C, ld=None, perm=None):
            if taps is None:
                return
            ld = C if ld is None else ld
            v = t[: rows * ld].view(rows, ld)[:, :C]
            if perm is not None:
                v = v[perm]
            taps[name] = v.clone()

        if images.dtype == torch.uint8:      # uint8 pixels: ToTensor + Normalize + the input QuantAct as a 3 x 256 table (engine.py)
            if self.input_lut is None:
                self.set_input_normalisation()
            _lib.call("ivit_quantize_patchify_u8_i8", _lib.ptr(images), _lib.ptr(ws["a0"]), 64, B, 3, IMG_SIZE, PATCH,
                      _lib.ptr(self.input_lut), st)
        else:
            _lib.call("ivit_quantize_patchify_ld_f32_i8", _lib.ptr(images), _lib.ptr(ws["a0"]), 64, B, 3, IMG_SIZE, PATCH,
                      self.inv_s0, st)
        self._gemm(ws["a0"], 64, self.patch, ws["pe"], C0, M, st)
        tap("patch_embed.qact_before_norm", ws["pe"], M, C0)
        ln = self.patch_ln
        if ln["remap"] is not None:
            _lib.call("ivit_layernorm_i8_compat", _lib.ptr(ws["pe"]), C0, M, C0, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(ln["remap"]), _lib.ptr(ln["phi"]), _lib.ptr(ws["pn"]), C0,
                      (M // B) << 8, st)   # IVIT_LN_OUTER_MEAN(L): the reference reduces over the transposed view of layers_quant.py:198
        else:
            _lib.call("ivit_layernorm_i8", _lib.ptr(ws["pe"]), C0, M, C0, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(ws["pn"]), C0, st)
        tap("patch_embed.qact", ws["pn"], M, C0)
        x, x2 = ws["x"], ws["x2"]
        _lib.call("ivit_requant_i8_i16", _lib.ptr(ws["pn"]), self.stem_me[0], self.stem_me[1], _lib.ptr(x), M * C0, st)
        tap("qact1", x, M, C0)

        for li, stg in enumerate(self.stages):
            H, W, C, nH = stg["H"], stg["W"], stg["C"], stg["nH"]
            M = B * H * W
            ld = _pad64(C)
            for bi, blk in enumerate(stg["blocks"]):
                p = f"layers.{li}.blocks.{bi}."
                win, shift = blk["win"], blk["shift"]
                N = win * win
                nwin = M // N
                perm = None
                if taps is not None:
                    perm = torch.from_numpy(window_row_map(B, H, W, win, shift)).to(self.dev)
                # every LayerNorm of stage 0 still sees the patch embedding's transposed layout in the reference: elementwise ops keep
                # the strides of layers_quant.py:198's view, and the residual QuantActs add `identity + x` with the identity (the
                # strided stream) as the first operand, whose layout torch then gives the sum; only the patch merging's cat makes the
                # stream contiguous.  Their float32 means run in torch's outer-reduction order (IVIT_LN_OUTER_MEAN)
                outer = H * W if li == 0 else 0
                self._ln16(x, M, C, blk["ln1"], ws["h"], ld, st, H, W, win, shift, outer=outer)
                tap(p + "qact1", ws["h"], M, C, ld, perm)
                q = blk["qkv"]
                qw, qlay = self._w(q, M)
                _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(ws["h"]), ld, qw, q["K"], _lib.ptr(q["b"]),
                          _lib.ptr(q["m"]), _lib.ptr(q["e"]), _lib.ptr(ws["qkv"]), N, nH, HEAD_DIM, M, 3 * C, q["K"], qlay, st)
                if taps is not None:   # reference layout [B_, N, 3C] (swin_quant.py:131-133)
                    hm = ws["qkv"][: 3 * M * C].view(3, nwin, nH, N, HEAD_DIM)
                    taps[p + "attn.qact1"] = hm.permute(1, 3, 0, 2, 4).reshape(nwin, N, 3 * C).clone()
                a = blk["attn"]
                fuse_proj = self.proj_fused and taps is None
                if a["band"] is not None:
                    _lib.call("ivit_window_attention_i8_band", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), ld, _lib.ptr(a["bias"]),
                              _lib.ptr(a["region"]), nwin, a["nW"], nH, N, HEAD_DIM, a["ms"][0], a["ms"][1], a["mb"][0], a["mb"][1],
                              a["s_attn"], a["mo"][0], a["mo"][1], _lib.ptr(a["band"]), a["band_w"], int(a["band"].shape[0]), H, W,
                              win if fuse_proj else 0, shift, st)
                elif fuse_proj:
                    # the attention output goes straight to its image rows (window reverse + roll back in the store address):
                    # attn.proj + attn.qact4 + the residual QuantAct qact2 are then ONE GEMM, in place on the residual stream
                    _lib.call("ivit_window_attention_i8_unwindow", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), ld, _lib.ptr(a["bias"]),
                              _lib.ptr(a["region"]), a["mask_value"], nwin, a["nW"], nH, N, HEAD_DIM, a["ms"][0], a["ms"][1],
                              a["mb"][0], a["mb"][1], a["s_attn"], a["mo"][0], a["mo"][1], _lib.ptr(a["phi"]), _lib.ptr(a["phim"]),
                              H, W, win, shift, st)
                else:
                    _lib.call("ivit_window_attention_i8_compat", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), ld, _lib.ptr(a["bias"]),
                              _lib.ptr(a["region"]), a["mask_value"], nwin, a["nW"], nH, N, HEAD_DIM, a["ms"][0], a["ms"][1], a["mb"][0],
                              a["mb"][1], a["s_attn"], a["mo"][0], a["mo"][1], _lib.ptr(a["phi"]), _lib.ptr(a["phim"]), st)
                tap(p + "attn.qact3", ws["ao"], M, C, ld)
                pj = blk["proj"]
                r = blk["res1"]
                if fuse_proj:
                    w_, lay = (_lib.ptr(pj["Wf"]), pj["Wf_bit"]) if (M >= 2048 and pj.get("Wf") is not None) else (_lib.ptr(pj["W"]), 0)
                    _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(ws["ao"]), ld, w_, pj["K"], _lib.ptr(pj["b"]),
                              _lib.ptr(pj["m"]), _lib.ptr(pj["e"]), _lib.ptr(x), C, r[0], r[1], r[2], r[3], _lib.ptr(x2), C, M, C,
                              pj["K"], lay, st)
                elif self.proj_i16:
                    # attn.proj + the 16-bit attn.qact4 in the GEMM epilogue (int16 [M, C] in window order, half the bytes of
                    # raw accumulators), then window reverse / un-shift + the residual QuantAct
                    _lib.call("ivit_gemm_i8_requant_i16", _lib.ptr(ws["ao"]), ld, _lib.ptr(pj["W"]), pj["K"], _lib.ptr(pj["b"]),
                              _lib.ptr(pj["m"]), _lib.ptr(pj["e"]), _lib.ptr(ws["acc"]), C, M, C, pj["K"], st)
                    _lib.call("ivit_residual_requant_i16", _lib.ptr(ws["acc"]), 16, None, None,
                              r[0], r[1], _lib.ptr(x), r[2], r[3], _lib.ptr(x2), M, C, H, W, win, shift, st)
                else:      # A/B: raw int32 accumulators, attn.qact4 inside the residual kernel
                    _lib.call("ivit_gemm_i8_i32", _lib.ptr(ws["ao"]), ld, _lib.ptr(pj["W"]), pj["K"], _lib.ptr(pj["b"]),
                              _lib.ptr(ws["acc"]), C, M, C, pj["K"], st)
                    _lib.call("ivit_residual_requant_i16", _lib.ptr(ws["acc"]), 32, _lib.ptr(pj["m"]), _lib.ptr(pj["e"]),
                              r[0], r[1], _lib.ptr(x), r[2], r[3], _lib.ptr(x2), M, C, H, W, win, shift, st)
                tap(p + "qact2", x2, M, C)
                self._ln16(x2, M, C, blk["ln2"], ws["h"], ld, st, outer=outer)
                tap(p + "qact3", ws["h"], M, C, ld)
                self._gemm(ws["h"], ld, blk["fc1"], ws["f1"], 4 * C, M, st)
                tap(p + "mlp.qact_gelu", ws["f1"], M, 4 * C)
                _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(ws["f1"]), 4 * C, M, 4 * C, _lib.ptr(blk["gelu_lut"]),
                          _lib.ptr(ws["g"]), 4 * C, st)
                tap(p + "mlp.qact1", ws["g"], M, 4 * C)
                r = blk["res2"]
                f2 = blk["fc2"]
                if taps is not None:   # tests: the 8-bit mlp.qact2 tensor is not materialised on the fused path below
                    self._gemm(ws["g"], 4 * C, f2, ws["f2"], C, M, st)
                    tap(p + "mlp.qact2", ws["f2"], M, C)
                # mlp.fc2 + mlp.qact2 + the 16-bit residual QuantAct qact4 in one kernel (swin_quant.py:297-299)
                f2w, f2lay = self._w(f2, M)
                _lib.call("ivit_gemm_i8_requant_residual_i16_ex", _lib.ptr(ws["g"]), 4 * C, f2w, f2["K"],
                          _lib.ptr(f2["b"]), _lib.ptr(f2["m"]), _lib.ptr(f2["e"]), _lib.ptr(x2), C, r[0], r[1], r[2], r[3],
                          _lib.ptr(x), C, M, f2["N"], f2["K"], f2lay, st)
                tap(p + "qact4", x, M, C)
            dn = stg["down"]
            if dn is not None:
                p = f"layers.{li}.downsample."
                _lib.call("ivit_patch_merge_i16", _lib.ptr(x), _lib.ptr(ws["xm"]), B, H, W, C, st)
                M4 = M // 4
                self._ln16(ws["xm"], M4, 4 * C, dn["ln"], ws["hm"], 4 * C, st)
                tap(p + "qact1", ws["hm"], M4, 4 * C)
                self._gemm(ws["hm"], 4 * C, dn["red"], ws["red"], 2 * C, M4, st)
                tap(p + "qact2", ws["red"], M4, 2 * C)
                _lib.call("ivit_requant_i8_i16", _lib.ptr(ws["red"]), IDENT[0], IDENT[1], _lib.ptr(x), M4 * 2 * C, st)

        C, T = self.C_last, self.T_last
        self._ln16(x, B * T, C, self.ln_f, ws["hN"], C, st)
        tap("qact2", ws["hN"], B * T, C)
        _lib.call("ivit_avgpool_requant_i8", _lib.ptr(ws["hN"]), _lib.ptr(ws["pooled"]), B, T, C, self.pool_me[0],
                  self.pool_me[1], st)
        tap("qact3", ws["pooled"], B, C)
        hd = self.head
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(ws["pooled"]), C, _lib.ptr(hd["W"]), hd["K"], _lib.ptr(hd["b"]),
                  _lib.ptr(ws["logits"]), hd["N"], B, hd["N"], hd["K"], st)
        _lib.call("ivit_head_argmax", _lib.ptr(ws["logits"]), _lib.ptr(self.head_scale), B, hd["N"],
                  _lib.ptr(ws["logits_f"]), _lib.ptr(ws["top1"]), st)
        nc = self.num_classes
        return ws["logits"][:B, :nc], ws["logits_f"][:B, :nc], ws["top1"][:B]

    __call__ = forward
