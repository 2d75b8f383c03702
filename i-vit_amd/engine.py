"""Integer-only DeiT/ViT forward on MI355X: int8 activations end to end, every operator a
hand-written HIP kernel reached through the C ABI (include/ivit_hip.h).

Dataflow = the reference's frozen-model forward
(/root/reference/models/vit_quant.py:285-312, Attention :61-90, Block :142-155,
 /root/reference/models/layers_quant.py:145-154, 191-203), restated on integers:

  images f32 --quantize+im2col--> int8 [B*196, 768]
    --GEMM(patch_embed.proj)+requant(patch_embed.qact)--> int8 [B*196, C]
    --cls/pos assemble (qact_pos, qact1)--> x int8 [B*197, C]
  12 x { LN+requant -> GEMM qkv (+requant, head-major) -> fused attention
         -> GEMM proj (+requant +residual requant) -> LN+requant -> GEMM fc1 (+requant)
         -> ShiftGELU table gather (+requant) -> GEMM fc2 (+requant +residual requant) }
    --LN(cls rows)+requant--> GEMM head --> INT32 logits --scale+argmax--> top-1

There is no fallback path: a missing libivit_hip.so or a kernel error raises.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from .graph import GraphReplay
from .prepare import (IMAGENET_MEAN, IMAGENET_STD, LayerNormParams, dyadic, f32, input_lut_u8, markstein_division_ok, pad_head, phi_is_identity, phi_tables, quant_sym,
                      requant_host, shiftexp2d, shiftexp_band)


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


class IntViTEngine(GraphReplay):
    # fragment-packed weights in the 16x16x64 MFMA order (False / IVIT_FRAGS16=0: the 32x32x32 order everywhere; A/B, tests)
    frags16 = os.environ.get("IVIT_FRAGS16", "1") != "0"

    def __init__(self, float_state=None, ranges=None, embed_dim: int = 768, depth: int = 12, num_heads: int = 12,
                 device="cuda:0", max_batch: int = 256, source=None, family: str = "ivit", stream_bits: int = 8,
                 softmax_bits: int = 8, pos_bits: int = 8, img_size: int = 224, patch_size: int = 16):
        """float_state: name -> float32 array (the reference's state_dict names);
        ranges: QuantAct name -> (x_min, x_max) of the frozen model.  Alternatively `source`: any object with the
        FloatSource interface of export.py (e.g. export.ExportSource: integer parameters + scale table, no floats)."""
        self.C, self.D, self.H = embed_dim, depth, num_heads
        # geometry (vit_quant.py:158-166: img_size, patch_size): square images, non-overlapping patches.  The patch GEMM steps K in
        # 64-byte slabs and the fused attention kernel holds a whole Shiftmax row of at most 207 keys in four lanes
        self.IMG, self.P = int(img_size), int(patch_size)
        if self.IMG % self.P or (3 * self.P * self.P) % 64 or (self.IMG // self.P) ** 2 + 1 > 207:
            raise ValueError(f"geometry {self.IMG} / {self.P}: needs img_size % patch_size == 0, 3 * patch_size^2 % 64 == 0 and at most 207 tokens")
        self.NP = (self.IMG // self.P) ** 2
        self.T = self.NP + 1
        # operator family of LayerNorm / Softmax / GELU: "ivit" (I-ViT: IVITIntLayerNorm, Shiftmax, ShiftGELU) or "ibert" (the
        # fork's default, ibert_modules.py: literal float32 sequences on fl(q * s), any activation scale)
        if family not in ("ivit", "ibert"):
            raise ValueError(f"operator family {family!r}: the fused engine implements 'ivit' and 'ibert'")
        self.family = family
        # width of the residual stream and of the QuantActs that feed it (vit_quant.py:180-187): 8, or 16 = patch_embed_bw,
        # block_input_bw, attention_out_bw, mlp_out_bw, norm2_in_bw, att_block_out_bw all 16 (softmax_bw, pos_encoding_bw 8): the
        # GEMM operands stay int8, the stream and the projection / fc2 outputs are int16 (the kernels of the Swin engine)
        if stream_bits not in (8, 16):
            raise ValueError("stream_bits must be 8 or 16")
        self.stream_bits = sb = stream_bits
        # softmax_bw / pos_encoding_bw (vit_quant.py:181, 184) may be 16 on the 16-bit-stream path ('--bitwidth 16' sets all eight)
        if softmax_bits not in (8, 16) or pos_bits not in (8, 16) or (sb == 8 and (softmax_bits, pos_bits) != (8, 8)):
            raise ValueError("softmax_bits / pos_bits: 8, or 16 together with stream_bits = 16")
        self.softmax_bits, self.pos_bits = softmax_bits, pos_bits
        self.hd = embed_dim // num_heads
        if self.hd != 64:
            raise ValueError("fused attention kernel supports head_dim 64 only")
        self.dev = torch.device(device)
        self.max_batch = max_batch
        _lib.lib()  # fail loudly now if the HIP library is absent
        if source is None:
            from .export import FloatSource
            source = FloatSource({k: _np(v) for k, v in float_state.items()}, ranges)
        C, H, hd = self.C, self.H, self.hd
        T = self.T
        s = source.act_scale

        def dev(a, dtype=None):
            t = torch.from_numpy(np.ascontiguousarray(a))
            if dtype is not None:
                t = t.to(dtype)
            return t.to(self.dev)

        def lin_dev(lp, s_out):
            m, e = lp.requant_to(s_out)
            return dict(W=dev(lp.W8), b=dev(lp.b32), m=dev(m.view(np.int32)), e=dev(e), K=lp.K, N=lp.W8.shape[0], Wb=None)

        def phi_dev(s_in):
            """natural (non power-of-two) scale of an operator's input: the reference's operator sees phi(q) = fl(fl(q*s)/s),
            not q (prepare.py).  -> (remap int8[256], phi f32[256]) on the device, or (None, None) when phi is the identity"""
            t = phi_tables(s_in)
            if t is None:
                return None, None
            self.natural_sites += 1
            return dev(t[0]), dev(t[1])

        def ln_dev(prefix, s_out, s_in):
            lp = source.layernorm(prefix, s_out)
            if sb == 16 and family == "ibert":
                try:
                    shift = float(np.asarray(source.tensor(prefix + ".shift")).reshape(-1)[0])
                except KeyError:
                    shift = 0.0
                return dict(kind="ib16", bias=dev(lp.bias_int), s=dev(lp.s_ln), m=dev(lp.m.view(np.int32)), e=dev(lp.e),
                            s_in=float(s_in), shift_pow2=float(2.0 ** shift), fast_div=int(markstein_division_ok(s_in, 16)))
            if sb == 16:
                # I-LayerNorm on the 16-bit stream (csrc/swin.hip); natural input scale: the literal / Markstein-quotient forms
                d = dict(kind="i16", bias=dev(lp.bias_int), s=dev(lp.s_ln), m=dev(lp.m.view(np.int32)), e=dev(lp.e), s_in=None, fast_div=0)
                if not phi_is_identity(s_in, 16):
                    self.natural_sites += 1
                    d.update(s_in=float(s_in), fast_div=int(markstein_division_ok(s_in, 16)))
                return d
            if family == "ibert":
                # IBERTIntLayerNorm has the same per-channel constants (bias_int, s_out = sqrt(C) / 2^30 * gamma,
                # ibert_modules.py:145-153) and an overflow shift buffer (:134-137); the kernel works on fl(q * s_in) literally
                try:
                    shift = float(np.asarray(source.tensor(prefix + ".shift")).reshape(-1)[0])
                except KeyError:
                    shift = 0.0
                return dict(kind="ibert", bias=dev(lp.bias_int), s=dev(lp.s_ln), m=dev(lp.m.view(np.int32)), e=dev(lp.e),
                            s_in=float(s_in), shift_pow2=float(2.0 ** shift), remap=None, phi=None)
            remap, phi = phi_dev(s_in)
            return dict(bias=dev(lp.bias_int), s=dev(lp.s_ln), m=dev(lp.m.view(np.int32)), e=dev(lp.e), remap=remap, phi=phi)

        def scalar_me(pre, z):
            m, e = dyadic(pre, z)
            return int(m[0]), int(e[0])

        def ranges_of(name):
            if ranges is None or name not in ranges:
                raise KeyError(f"the 'ibert' engine needs the range of {name} (the softmax's internal 16-bit QuantAct)")
            return float(ranges[name][0]), float(ranges[name][1])

        self.natural_sites = 0     # operators whose input scale is not a power of two (compat kernels / tables)
        # ---- stem
        s0 = s("qact_input")
        self.inv_s0 = float(f32(1.0) / s0)
        self.s0 = float(s0)
        self.input_lut = None
        pe = source.linear("patch_embed.proj", s0)
        s_pe = s("patch_embed.qact", sb)
        self.patch = lin_dev(pe, s_pe)
        s_pos, s_x = s("qact_pos", pos_bits), s("qact1", sb)
        m1, e1 = dyadic(s_pe, s_x)
        m2, e2 = dyadic(s_pos, s_x)
        kpos = quant_sym(source.tensor("pos_embed").reshape(T, C), s_pos, pos_bits)
        pos_add = requant_host(kpos, m2[0], e2[0])                       # RNE(k_pos * m2 / 2^e2)
        z_cls = np.rint((source.tensor("cls_token").reshape(C) / s_pe).astype(f32))   # quant_utils.py:220 on the raw cls row
        qlim = 2 ** (sb - 1)
        cls_row = np.clip(requant_host(z_cls, m1[0], e1[0]) + pos_add[0], -qlim, qlim - 1)
        if sb == 16:
            assert np.abs(pos_add).max() < 2 ** 31
            self.pos_add = dev(pos_add.astype(np.int32))
            self.cls_row = dev(cls_row.astype(np.int16))
        else:
            assert np.abs(pos_add).max() < 32768
            self.pos_add = dev(pos_add.astype(np.int16))
            self.cls_row = dev(cls_row.astype(np.int8))
        self.embed_me = (int(m1[0]), int(e1[0]))

        # ---- blocks
        self.blocks = []
        for i in range(depth):
            p = f"blocks.{i}."
            blk = {}
            s_q1 = s(p + "qact1")
            blk["ln1"] = ln_dev(p + "norm1", s_q1, s_x)
            s_a1 = s(p + "attn.qact1")
            blk["qkv"] = lin_dev(source.linear(p + "attn.qkv", s_q1), s_a1)
            s_S = f32(f32(s_a1 * s_a1) * f32(hd ** -0.5))                 # vit_quant.py:72-75
            s_at = s(p + "attn.qact_attn1")
            s_pv = f32(f32(1.0 / 2 ** (softmax_bits - 1)) * s_a1)         # Shiftmax scale 2^-(bits-1) (:176) x value scale
            s_a2 = s(p + "attn.qact2")
            blk["attn"] = dict(ms=scalar_me(s_S, s_at), s_attn=float(s_at), mo=scalar_me(s_pv, s_a2), exp2d=None, band=None, band_w=0)
            if family == "ibert":
                # IBERTIntSoftmax (output_bit 8, scale 2 / 2^8 = 2^-7 like Shiftmax's): exp_int after its internal 16-bit
                # QuantAct, tabulated over (row max, q) with the reference's float32 sequence (csrc/ibert.hip)
                from .quantization_utils.ibert_modules import softmax_constants
                lo, hi = ranges_of(p + "attn.int_softmax.act")
                x0i, bi, ci, exp_sf, act_sf, ma, ea = softmax_constants(s_at, lo, hi)
                tab = torch.empty(65536, dtype=torch.float32, device=self.dev)
                _lib.call("ivit_ibert_softmax_build_table", float(s_at), x0i, bi, ci, float(exp_sf), float(act_sf), ma, ea,
                          _lib.ptr(tab), self._stream())
                blk["attn"]["ib_table"] = tab
                # band form (LDS path) when the exponent saturates within 128 steps of the row maximum
                band, bw = shiftexp_band(tab.cpu().numpy().view(np.uint32).reshape(256, 256))
                if bw and bw <= 128:
                    blk["attn"].update(band=dev(band.view(np.float32)), band_w=bw)
            elif phi_tables(s_at) is not None:       # Shiftmax on phi(q): exponent tabulated over (row max, q)
                self.natural_sites += 1
                tab = shiftexp2d(s_at)
                band, bw = shiftexp_band(tab)
                if bw and bw <= 128:               # band rows staged in LDS per query tile (34 KB per workgroup at width 128)
                    blk["attn"].update(band=dev(band.view(np.int32)), band_w=bw)
                else:                              # very fine input scale: full-table gather
                    blk["attn"]["exp2d"] = dev(tab.view(np.int32))
            s_a3 = s(p + "attn.qact3", sb)
            blk["proj"] = lin_dev(source.linear(p + "attn.proj", s_a2), s_a3)
            s_b2 = s(p + "qact2", sb)
            blk["res1"] = scalar_me(s_a3, s_b2) + scalar_me(s_x, s_b2)
            s_b3 = s(p + "qact3")
            blk["ln2"] = ln_dev(p + "norm2", s_b3, s_b2)
            s_g = s(p + "mlp.qact_gelu")
            blk["fc1"] = lin_dev(source.linear(p + "mlp.fc1", s_b3), s_g)
            s_m1 = s(p + "mlp.qact1")
            lut = torch.empty(65536, dtype=torch.int8, device=self.dev)
            if family == "ibert":
                # IBERTIntGELU + mlp.qact1 depend on q alone: one 256-entry map, replicated over the table's row-max axis.  Its
                # output scale is negative (ibert_modules.py:213, 232): requant(z, s) == requant(-z, -s) (quant_modules.QuantAct)
                from .quantization_utils.ibert_modules import gelu_constants
                gb, gc, gsh, gso = gelu_constants(s_g)
                mg, eg = scalar_me(abs(f32(gso)), s_m1)
                _lib.call("ivit_ibert_gelu_build_lut", float(s_g), gb, gc, gsh, float(gso), mg, eg, _lib.ptr(lut), self._stream())
            else:
                s_go = f32(s_g * f32(1.0 / 128.0))                             # ivit_modules.py:121,124
                mg, eg = scalar_me(s_go, s_m1)
                g_remap, _ = phi_dev(s_g)              # ShiftGELU sees trunc(phi(q)) (ivit_modules.py:106-107)
                _lib.call("ivit_shiftgelu_build_lut_ex", float(s_g), mg, eg, _lib.ptr(g_remap), _lib.ptr(lut), self._stream())
            blk["gelu_lut"] = lut
            s_m2 = s(p + "mlp.qact2", sb)
            blk["fc2"] = lin_dev(source.linear(p + "mlp.fc2", s_m1), s_m2)
            s_b4 = s(p + "qact4", sb)
            blk["res2"] = scalar_me(s_m2, s_b4) + scalar_me(s_b2, s_b4)
            s_x = s_b4
            self.blocks.append(blk)

        # ---- tail
        s_q2 = s("qact2")
        self.ln_f = ln_dev("norm", s_q2, s_x)
        head = source.linear("head", s_q2)
        hW, hb, hs, self.num_classes = pad_head(head.W8, head.b32, head.s_acc)    # any class count (num_classes=... of the factory)
        self.head = dict(W=dev(hW), b=dev(hb), K=head.K, N=hW.shape[0])
        self.head_scale = dev(hs)
        self.int8_weight_bytes = sum(int(b[k]["W"].numel()) for b in self.blocks for k in ("qkv", "proj", "fc1", "fc2")) \
            + int(self.patch["W"].numel()) + int(self.head["W"].numel())
        # block-layout copies of the GEMM weights (include/ivit_hip.h IVIT_LAYOUT_BLOCKS): the persistent GEMM then reads
        # 1 KB contiguous per LDS-DMA instruction instead of 16 half cache lines
        for lin in [self.patch] + [b[k] for b in self.blocks for k in ("qkv", "proj", "fc1", "fc2")]:
            if lin["K"] % 64 == 0 and lin["N"] % 16 == 0:
                lin["Wb"] = torch.empty_like(lin["W"])
                _lib.call("ivit_tile_operand_i8", _lib.ptr(lin["W"]), lin["K"], lin["N"], lin["K"], _lib.ptr(lin["Wb"]),
                          self._stream())
        # MFMA-fragment copies: the weights-in-registers GEMM, 14-21 % faster than the LDS-DMA kernel on block-layout weights; needs
        # K % 192 == 0, and its 256-channel tiles must not waste more than an eighth of their columns.  IVIT_W_FRAGS16 (the
        # v_mfma_i32_16x16x64_i8 form: the chip holds a higher clock on that shape, fc1 -7 %, fc2 -5 %) wherever the epilogue writes
        # int8; the 16-bit-stream epilogue of proj / fc2 exists for the 32x32x32 form (IVIT_W_FRAGS) only
        wide = {id(b[k]) for b in self.blocks for k in ("proj", "fc2")} if self.stream_bits == 16 else set()
        for lin in [self.patch] + [b[k] for b in self.blocks for k in ("qkv", "proj", "fc1", "fc2")]:
            N, K = lin["N"], lin["K"]
            lin["Wf"] = None
            lin["Wf_bit"] = 8 if id(lin) in wide or not self.frags16 else 16
            # (the 16x16x64 form has 128-channel work items for the widths 256-channel tiles fit badly, round 4: any N % 64 == 0)
            if K % 192 == 0 and N % 64 == 0 and N >= 128 and (lin["Wf_bit"] == 16 or (N + 255) // 256 * 256 * 8 <= N * 9):
                lin["Wf"] = torch.empty((N + 63) // 64 * 64 * K, dtype=torch.int8, device=self.dev)
                _lib.call("ivit_pack_weight_frags_i8" if lin["Wf_bit"] == 8 else "ivit_pack_weight_frags16_i8", _lib.ptr(lin["W"]), K, N, K,
                          _lib.ptr(lin["Wf"]), self._stream())
        self.weight_frags = True      # False: block-layout weights through the LDS-DMA kernel (A/B timing)
        self.block_operands = True    # False: row-major activations / weights everywhere (tests, A/B timing)
        # which producers write their output (a GEMM A operand) in the block layout.  Measured per producer / consumer pair
        # (DESIGN.md section 5): the GEMM gains 4-6 % from a block-layout A, the producer pays for 64-byte row segments
        self.block_a = {"ln": True, "attn": True, "gelu": True}
        self.gelu_in_place = True     # GELU overwrites the fc1 output (same layout on both sides)
        self.fuse_res16 = True        # 16-bit stream: projection / fc2 + both QuantActs in one kernel (False: A/B, tests)
        self.fuse_ibert_gelu = True   # family "ibert": GELU + mlp.qact1 as a byte map in the fc1 epilogue (False: A/B, tests)
        self.probe = None
        self._alloc(max_batch)
        self._compact(True)
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return _lib.stream_ptr()

    def _alloc(self, B):
        C, T = self.C, self.T
        M = B * T
        M16 = (M + 15) // 16 * 16   # block-layout operands pad their rows to a multiple of 16
        i8 = dict(dtype=torch.int8, device=self.dev)
        self.ws = dict(
            a0=torch.empty(B * self.NP, 3 * self.P * self.P, **i8),
            pe=torch.empty(B * self.NP, C, **i8),
            x=torch.empty(M, C, **i8), x2=torch.empty(M, C, **i8), h=torch.empty(M16, C, **i8),
            qkv=torch.empty(3 * M * C, **i8), ao=torch.empty(M16, C, **i8),
            f1=torch.empty(M16, 4 * C, **i8), g=torch.empty(M16, 4 * C, **i8), untile=torch.empty(M, 4 * C, **i8),
            cls=torch.empty(B, C, **i8),
            **({} if self.stream_bits == 8 else dict(
                pe16=torch.empty(B * self.NP, C, dtype=torch.int16, device=self.dev),
                x16=torch.empty(M, C, dtype=torch.int16, device=self.dev), y16=torch.empty(M, C, dtype=torch.int16, device=self.dev),
                k16=torch.empty(M, C, dtype=torch.int16, device=self.dev),
                cls16=torch.empty(B, C, dtype=torch.int16, device=self.dev))),
            logits=torch.empty(B, self.head["N"], dtype=torch.int32, device=self.dev),
            logits_f=torch.empty(B, self.head["N"], dtype=torch.float32, device=self.dev),
            top1=torch.empty(B, dtype=torch.int32, device=self.dev),
        )

    def _w(self, lin, blocks):
        """(weight pointer, layout bit) -- the block-layout copy when the call goes to the persistent kernel"""
        if blocks and self.weight_frags and lin.get("Wf") is not None:
            return _lib.ptr(lin["Wf"]), lin["Wf_bit"]
        if blocks and lin["Wb"] is not None:
            return _lib.ptr(lin["Wb"]), 2
        return _lib.ptr(lin["W"]), 0

    def _compact(self, on=True):
        """Alias workspaces whose lifetimes do not overlap, so that one layer touches ~230 MB instead of ~430 MB at batch
        256 (the Infinity Cache holds 256 MB): the residual QuantActs run in place (x2 = x; the epilogue's thread reads a
        residual chunk and writes the same chunk), the attention output reuses the LayerNorm buffer (consumed by the qkv
        GEMM before attention starts), q/k/v live inside the fc1 / GELU buffer (dead before fc1 writes it)."""
        ws = self.ws
        if "_own" not in ws:
            ws["_own"] = {k: ws[k] for k in ("x2", "ao", "qkv")}
        if on:
            ws["x2"] = ws["x"]
            ws["ao"] = ws["h"]
            ws["qkv"] = ws["f1"].view(-1)[: ws["_own"]["qkv"].numel()]
        else:
            ws.update(ws["_own"])

    def _gemm(self, A, lda, lin, out, ldo, M, st, a_blocks=False, blocks=False, out_blocks=False):
        w, lay = self._w(lin, blocks)
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(A), lda, w, lin["K"], _lib.ptr(lin["b"]),
                  _lib.ptr(lin["m"]), _lib.ptr(lin["e"]), _lib.ptr(out), ldo, M, lin["N"], lin["K"],
                  lay | int(a_blocks) | (4 if out_blocks else 0), st)

    def _gemm_res(self, A, lda, lin, res, me4, out, M, st, blocks=False, a_blocks=False):
        C = self.C
        w, lay = self._w(lin, blocks)
        lay |= int(a_blocks)
        probe = self.probe
        if probe is not None:
            # bench.py's separate instrumented pass (never inside its timed region): HIP events around the dominant kernel
            probe.begin("gemm_resid", st)
        _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(A), lda, w, lin["K"],
                  _lib.ptr(lin["b"]), _lib.ptr(lin["m"]), _lib.ptr(lin["e"]), _lib.ptr(res), C,
                  me4[0], me4[1], me4[2], me4[3], _lib.ptr(out), C, M, lin["N"], lin["K"], lay, st)
        if probe is not None:
            probe.end("gemm_resid", st, (M, lin["N"], lin["K"]))

    def _ln(self, x, ldx, rows, ln, out, st, blocks=False):
        C = self.C
        if ln.get("kind") == "ibert":
            _lib.call("ivit_ibert_layernorm_i8", _lib.ptr(x), ldx, rows, C, ln["s_in"], _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      ln["shift_pow2"], _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), C, int(blocks), st)
            return
        if ln["remap"] is not None:
            _lib.call("ivit_layernorm_i8_compat", _lib.ptr(x), ldx, rows, C, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(ln["remap"]), _lib.ptr(ln["phi"]), _lib.ptr(out), C,
                      int(blocks), st)
            return
        _lib.call("ivit_layernorm_i8_ex", _lib.ptr(x), ldx, rows, C, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                  _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), C, int(blocks), st)

    def set_input_normalisation(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        """uint8 input: the (mean, std) of the Normalize transform in front of the model (default: ImageNet's).  forward() then
        accepts uint8 [B,3,224,224] pixel tensors -- a quarter of the bytes of the float32 input -- and quantises them through a
        3 x 256 table that holds the float pipeline's result per (channel, pixel value): same integers as the float path."""
        self.input_lut = torch.from_numpy(input_lut_u8(self.s0, mean, std)).to(self.dev)

    def _patchify(self, images, B, st):
        ws = self.ws
        if images.dtype == torch.uint8:
            if self.input_lut is None:
                self.set_input_normalisation()
            _lib.call("ivit_quantize_patchify_u8_i8", _lib.ptr(images), _lib.ptr(ws["a0"]), 3 * self.P * self.P, B, 3, self.IMG, self.P,
                      _lib.ptr(self.input_lut), st)
        else:
            _lib.call("ivit_quantize_patchify_f32_i8", _lib.ptr(images), _lib.ptr(ws["a0"]), B, 3, self.IMG, self.P, self.inv_s0, st)

    # ------------------------------------------------------------------ forward
    def forward(self, images: torch.Tensor, taps: dict | None = None):
        """images: float32 [B,3,224,224] on the engine's device.  Returns (logits_int32 [B,classes],
        logits_f32 [B,classes], top1 int32 [B]) -- views of the engine's workspace, valid until the
        next call.  `taps` (debug/tests) receives clones of intermediate int8 tensors."""
        assert images.is_cuda and images.dtype in (torch.float32, torch.uint8) and images.is_contiguous()
        B = images.shape[0]
        assert images.shape[1:] == (3, self.IMG, self.IMG) and 0 < B <= self.max_batch
        if self.stream_bits == 16:
            if taps is not None:
                raise NotImplementedError("taps are not recorded on the 16-bit-stream path")
            return self._forward16(images)
        C, H, hd, T = self.C, self.H, self.hd, self.T
        M = B * T
        ws = self.ws
        st = self._stream()

        # GEMM operands in the block layout whenever the calls go to the persistent kernel (M >= 2048; N >= 128 always)
        blk_l = bool(self.block_operands) and M >= 2048 and C % 64 == 0    # weights (always) and, per producer, activations
        a_ln, a_at, a_ge = (blk_l and self.block_a[k] for k in ("ln", "attn", "gelu"))

        def tap(name, t, shape, blocks=False):
            if taps is not None:
                if blocks:   # back to rows for the caller
                    rows, K = int(np.prod(shape[:-1])), shape[-1]
                    _lib.call("ivit_untile_operand_i8", _lib.ptr(t), rows, K, _lib.ptr(ws["untile"]), K, st)
                    t = ws["untile"]
                taps[name] = t.reshape(-1)[: int(np.prod(shape))].view(shape).clone()

        self._patchify(images, B, st)
        self._gemm(ws["a0"], 3 * self.P * self.P, self.patch, ws["pe"], C, B * self.NP, st,
                   blocks=bool(self.block_operands) and B * self.NP >= 2048 and C >= 128)
        tap("patch_embed.qact", ws["pe"], (B, self.NP, C))
        _lib.call("ivit_embed_assemble_i8", _lib.ptr(ws["pe"]), _lib.ptr(self.pos_add), _lib.ptr(self.cls_row),
                  self.embed_me[0], self.embed_me[1], _lib.ptr(ws["x"]), B, T, C, st)
        tap("qact1", ws["x"], (B, T, C))
        x, x2 = ws["x"], ws["x2"]
        for i, blk in enumerate(self.blocks):
            p = f"blocks.{i}."
            self._ln(x, C, M, blk["ln1"], ws["h"], st, blocks=a_ln)
            tap(p + "qact1", ws["h"], (B, T, C), a_ln)
            q = blk["qkv"]
            qw, qlay = self._w(q, blk_l)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(ws["h"]), C, qw, q["K"], _lib.ptr(q["b"]),
                      _lib.ptr(q["m"]), _lib.ptr(q["e"]), _lib.ptr(ws["qkv"]), T, H, hd, M, 3 * C, C, qlay | int(a_ln), st)
            tap(p + "attn.qkv_headmajor", ws["qkv"], (3, B, H, T, hd))
            a = blk["attn"]
            if self.family == "ibert":
                _lib.call("ivit_attention_fused_i8_ibert", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), B, H, T, hd,
                          a["ms"][0], a["ms"][1], a["mo"][0], a["mo"][1], _lib.ptr(a["ib_table"]), _lib.ptr(a["band"]), a["band_w"],
                          int(a_at), st)
            else:
                _lib.call("ivit_attention_fused_i8_compat_band", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), B, H, T, hd,
                          a["ms"][0], a["ms"][1], a["s_attn"], a["mo"][0], a["mo"][1], _lib.ptr(a["exp2d"]), _lib.ptr(a["band"]),
                          a["band_w"], int(a_at), st)
            tap(p + "attn.qact2", ws["ao"], (B, T, C), a_at)
            self._gemm_res(ws["ao"], C, blk["proj"], x, blk["res1"], x2, M, st, blocks=blk_l, a_blocks=a_at)
            tap(p + "qact2", x2, (B, T, C))
            self._ln(x2, C, M, blk["ln2"], ws["h"], st, blocks=a_ln)
            tap(p + "qact3", ws["h"], (B, T, C), a_ln)
            # mlp.fc1 writes the block layout and GELU works IN PLACE on it: the 155 MB intermediate exists once, so the pair
            # (GELU output, fc2 operand) stays inside the 256 MB Infinity Cache (separate buffers: 310 MB; -0.18 ms / forward)
            f1 = blk["fc1"]
            if (self.family == "ibert" and self.fuse_ibert_gelu and taps is None and blk_l and self.weight_frags
                    and f1.get("Wf") is not None):
                # I-BERT GELU + mlp.qact1 is a map of the requantised byte alone (no row maximum): applied in the fc1 epilogue,
                # the GELU kernel and its pass over the 4C-wide intermediate disappear
                g_buf = ws["f1"]
                _lib.call("ivit_gemm_i8_requant_lut_ex", _lib.ptr(ws["h"]), C, _lib.ptr(f1["Wf"]), f1["K"], _lib.ptr(f1["b"]),
                          _lib.ptr(f1["m"]), _lib.ptr(f1["e"]), _lib.ptr(blk["gelu_lut"]), _lib.ptr(g_buf), 4 * C, M, f1["N"], f1["K"],
                          f1["Wf_bit"] | int(a_ln) | (4 if a_ge else 0), st)
            else:
                self._gemm(ws["h"], C, f1, ws["f1"], 4 * C, M, st, a_blocks=a_ln, blocks=blk_l, out_blocks=a_ge)
                tap(p + "mlp.qact_gelu", ws["f1"], (B, T, 4 * C), a_ge)
                g_buf = ws["f1"] if self.gelu_in_place else ws["g"]
                _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(ws["f1"]), 4 * C, M, 4 * C, _lib.ptr(blk["gelu_lut"]),
                          _lib.ptr(g_buf), 4 * C, 3 if a_ge else 0, st)
            tap(p + "mlp.qact1", g_buf, (B, T, 4 * C), a_ge)
            self._gemm_res(g_buf, 4 * C, blk["fc2"], x2, blk["res2"], x, M, st, blocks=blk_l, a_blocks=a_ge)
            tap(p + "qact4", x, (B, T, C))
        # final LayerNorm is row-wise and only the cls row is consumed (vit_quant.py:302-304)
        self._ln(x, T * C, B, self.ln_f, ws["cls"], st)
        tap("qact2", ws["cls"], (B, C))
        hd_ = self.head
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(ws["cls"]), C, _lib.ptr(hd_["W"]), hd_["K"], _lib.ptr(hd_["b"]),
                  _lib.ptr(ws["logits"]), hd_["N"], B, hd_["N"], C, st)
        _lib.call("ivit_head_argmax", _lib.ptr(ws["logits"]), _lib.ptr(self.head_scale), B, hd_["N"],
                  _lib.ptr(ws["logits_f"]), _lib.ptr(ws["top1"]), st)
        nc = self.num_classes
        return ws["logits"][:B, :nc], ws["logits_f"][:B, :nc], ws["top1"][:B]

    # ------------------------------------------------------------------ 16-bit residual stream
    def _ln16(self, x16, rows, ln, out, st):
        C = self.C
        if ln["kind"] == "ib16":
            _lib.call("ivit_ibert_layernorm_i16_i8_ex", _lib.ptr(x16), C, rows, C, ln["s_in"], _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]),
                      ln["shift_pow2"], _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), C, ln["fast_div"], st)
            return
        if ln["s_in"] is not None:
            _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(x16), rows, C, ln["s_in"], ln["fast_div"], _lib.ptr(ln["bias"]),
                      _lib.ptr(ln["s"]), _lib.ptr(ln["m"]), _lib.ptr(ln["e"]), _lib.ptr(out), C, 0, 0, 0, 0, st)
        else:
            _lib.call("ivit_layernorm_i16_i8", _lib.ptr(x16), rows, C, _lib.ptr(ln["bias"]), _lib.ptr(ln["s"]), _lib.ptr(ln["m"]),
                      _lib.ptr(ln["e"]), _lib.ptr(out), C, 0, 0, 0, 0, st)

    def _forward16(self, images: torch.Tensor):
        """stream_bits = 16: the same dataflow with an int16 residual stream.  LayerNorm reads int16 rows (csrc/swin.hip), the
        projection / fc2 GEMMs requantise their accumulators to 16 bits per channel (attn.qact3 / mlp.qact2 at 16 bits) and the
        residual QuantActs are the 16-bit two-operand kernel; qkv / fc1 / attention / GELU are the int8 kernels unchanged."""
        B = images.shape[0]
        ws, C, H, hd, T = self.ws, self.C, self.H, self.hd, self.T
        M = B * T
        st = self._stream()
        big = bool(self.block_operands) and M >= 2048 and C % 64 == 0
        self._patchify(images, B, st)
        pt = self.patch
        _lib.call("ivit_gemm_i8_requant_i16", _lib.ptr(ws["a0"]), 3 * self.P * self.P, _lib.ptr(pt["W"]), pt["K"], _lib.ptr(pt["b"]),
                  _lib.ptr(pt["m"]), _lib.ptr(pt["e"]), _lib.ptr(ws["pe16"]), C, B * self.NP, C, pt["K"], st)
        _lib.call("ivit_embed_assemble_i16", _lib.ptr(ws["pe16"]), _lib.ptr(self.pos_add), _lib.ptr(self.cls_row),
                  self.embed_me[0], self.embed_me[1], _lib.ptr(ws["x16"]), B, T, C, st)
        x = ws["x16"]
        inplace = bool(self.fuse_res16)

        def gemm_res16(A, lda, lin, r, res, out):
            # projection / fc2 + its 16-bit QuantAct + the block's residual QuantAct: one kernel in the weights-in-registers form
            if self.fuse_res16:
                frags = big and self.weight_frags and lin.get("Wf") is not None      # else: the 128 x 128-tile kernel, any shape
                probe = self.probe
                if probe is not None:     # bench.py's instrumented pass, as in _gemm_res
                    probe.begin("gemm_resid", st)
                _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(A), lda, _lib.ptr(lin["Wf"] if frags else lin["W"]), lin["K"],
                          _lib.ptr(lin["b"]), _lib.ptr(lin["m"]), _lib.ptr(lin["e"]), _lib.ptr(res), C, r[0], r[1], r[2], r[3], _lib.ptr(out),
                          C, M, C, lin["K"], lin["Wf_bit"] if frags else 0, st)
                if probe is not None:
                    probe.end("gemm_resid", st, (M, C, lin["K"]))
                return
            _lib.call("ivit_gemm_i8_requant_i16", _lib.ptr(A), lda, _lib.ptr(lin["W"]), lin["K"], _lib.ptr(lin["b"]),
                      _lib.ptr(lin["m"]), _lib.ptr(lin["e"]), _lib.ptr(ws["k16"]), C, M, C, lin["K"], st)
            _lib.call("ivit_residual_requant_i16", _lib.ptr(ws["k16"]), 16, None, None, r[0], r[1], _lib.ptr(res), r[2], r[3],
                      _lib.ptr(out), M, C, 0, 0, 0, 0, st)

        for blk in self.blocks:
            self._ln16(x, M, blk["ln1"], ws["h"], st)
            q = blk["qkv"]
            qw, qlay = self._w(q, big)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(ws["h"]), C, qw, q["K"], _lib.ptr(q["b"]),
                      _lib.ptr(q["m"]), _lib.ptr(q["e"]), _lib.ptr(ws["qkv"]), T, H, hd, M, 3 * C, C, qlay, st)
            a = blk["attn"]
            if self.family == "ibert":
                _lib.call("ivit_attention_fused_i8_ibert_wide", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), B, H, T, hd,
                          a["ms"][0], a["ms"][1], a["mo"][0], a["mo"][1], _lib.ptr(a["ib_table"]), _lib.ptr(a["band"]), a["band_w"],
                          self.softmax_bits, 0, st)
            else:
                _lib.call("ivit_attention_fused_i8_wide", _lib.ptr(ws["qkv"]), _lib.ptr(ws["ao"]), B, H, T, hd,
                          a["ms"][0], a["ms"][1], a["s_attn"], a["mo"][0], a["mo"][1], _lib.ptr(a["exp2d"]), _lib.ptr(a["band"]),
                          a["band_w"], self.softmax_bits, 0, st)
            y = x if inplace else ws["y16"]     # the fused epilogue's thread reads a residual chunk and writes the same chunk
            gemm_res16(ws["ao"], C, blk["proj"], blk["res1"], x, y)
            self._ln16(y, M, blk["ln2"], ws["h"], st)
            f1 = blk["fc1"]
            if self.family == "ibert" and self.fuse_ibert_gelu and big and self.weight_frags and f1.get("Wf") is not None:
                # I-BERT GELU + mlp.qact1 as a byte map in the fc1 epilogue (see forward)
                _lib.call("ivit_gemm_i8_requant_lut_ex", _lib.ptr(ws["h"]), C, _lib.ptr(f1["Wf"]), f1["K"], _lib.ptr(f1["b"]),
                          _lib.ptr(f1["m"]), _lib.ptr(f1["e"]), _lib.ptr(blk["gelu_lut"]), _lib.ptr(ws["f1"]), 4 * C, M, f1["N"], f1["K"],
                          f1["Wf_bit"], st)
            else:
                self._gemm(ws["h"], C, f1, ws["f1"], 4 * C, M, st, a_blocks=False, blocks=big, out_blocks=False)
                _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(ws["f1"]), 4 * C, M, 4 * C, _lib.ptr(blk["gelu_lut"]),
                          _lib.ptr(ws["f1"]), 4 * C, 0, st)
            gemm_res16(ws["f1"], 4 * C, blk["fc2"], blk["res2"], y, x)
        # final LayerNorm: only the cls rows are consumed (vit_quant.py:302-304); the int16 kernel wants dense rows
        ws["cls16"][:B].copy_(x.view(-1, T, C)[:B, 0])
        self._ln16(ws["cls16"], B, self.ln_f, ws["cls"], st)
        hd_ = self.head
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(ws["cls"]), C, _lib.ptr(hd_["W"]), hd_["K"], _lib.ptr(hd_["b"]),
                  _lib.ptr(ws["logits"]), hd_["N"], B, hd_["N"], C, st)
        _lib.call("ivit_head_argmax", _lib.ptr(ws["logits"]), _lib.ptr(self.head_scale), B, hd_["N"],
                  _lib.ptr(ws["logits_f"]), _lib.ptr(ws["top1"]), st)
        nc = self.num_classes
        return ws["logits"][:B, :nc], ws["logits_f"][:B, :nc], ws["top1"][:B]

    __call__ = forward
