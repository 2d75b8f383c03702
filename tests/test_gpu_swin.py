"""GPU parity for the Swin path (config 5): the extra HIP kernels against the CPU oracle, and the whole
IntSwinEngine against tests/golden/swin_tiny.npz (INT32 logits, top-1, CRC32 of the reference's taps)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ivit = pytest.importorskip("ivit_amd")
from ivit_amd import _lib, synth  # noqa: E402
from ivit_amd.checkpoint import load_fixture  # noqa: E402
from ivit_amd.prepare import LayerNormParams, dyadic  # noqa: E402
from ivit_amd.swin_engine import IntSwinEngine  # noqa: E402

DEV = "cuda:0"
_KEEP = []


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def st():
    return _lib.stream_ptr()


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.int32).tobytes())


def sme(pre, z):
    m, e = dyadic(np.float32(pre), np.float32(z))
    return int(m[0]), int(e[0])


def ome(pre, z):
    return orc.dyadic(np.float32(pre), np.float32(z))


# ----------------------------------------------------------------------------------- elementwise kernels
def test_requant_i8_i16():
    rng = np.random.default_rng(1)
    x = rng.integers(-128, 128, size=100003, dtype=np.int8)
    for pre, z in ((0.031, 0.00011), (1.0, 1.0), (0.5, 3.0)):
        m, e = sme(pre, z)
        out = torch.empty(x.size, dtype=torch.int16, device=DEV)
        _lib.call("ivit_requant_i8_i16", _lib.ptr(dev(x)), m, e, _lib.ptr(out), x.size, st())
        om, oe = ome(pre, z)
        ref = orc.requant(x.astype(np.int32).reshape(1, -1), om, oe, 16).reshape(-1)
        assert np.array_equal(out.cpu().numpy().astype(np.int32), ref)
    m, e = 1 << 30, 30
    out = torch.empty(x.size, dtype=torch.int16, device=DEV)
    _lib.call("ivit_requant_i8_i16", _lib.ptr(dev(x)), m, e, _lib.ptr(out), x.size, st())
    assert np.array_equal(out.cpu().numpy(), x.astype(np.int16))


@pytest.mark.parametrize("a_bits", [8, 16, 32])
@pytest.mark.parametrize("B,H,ws,shift,C", [(2, 14, 7, 3, 96), (3, 8, 4, 0, 64), (1, 7, 7, 0, 768), (2, 8, 4, 2, 36)])
def test_residual_requant_i16(a_bits, B, H, ws, shift, C):
    rng = np.random.default_rng(a_bits * 100 + H)
    rows = B * H * H
    res = rng.integers(-32768, 32768, size=(rows, C)).astype(np.int16)
    ma, mr = sme(0.013, 0.02), sme(0.0171, 0.02)
    oma, omr = ome(0.013, 0.02), ome(0.0171, 0.02)
    mp = ep = None
    if a_bits == 8:
        a = rng.integers(-128, 128, size=(rows, C)).astype(np.int8)
        k = a.astype(np.int32)
    elif a_bits == 16:
        a = rng.integers(-32768, 32768, size=(rows, C)).astype(np.int16)
        k = a.astype(np.int32)
    else:
        a = rng.integers(-200000, 200000, size=(rows, C)).astype(np.int32)
        pre = (rng.uniform(0.5, 1.0, size=C) * 2.0 ** rng.integers(-4, 1, size=C)).astype(np.float32)
        mpre, epre = dyadic(pre, np.float32(1.0))
        om, oe = orc.dyadic(pre, np.float32(1.0))
        k = orc.requant(a, om, oe, 16)
        mp, ep = dev(mpre.view(np.int32)), dev(epre)
    # reference: window_reverse + roll back of the window-ordered `k`, then the two-operand QuantAct
    kk = orc._win_reverse(k.reshape(-1, ws, ws, C), ws, H, H)
    if shift:
        kk = np.roll(kk, (shift, shift), axis=(1, 2))
    ref = orc.requant(np.ascontiguousarray(kk.reshape(rows, C)), oma[0], oma[1], 16, z2=res.astype(np.int32), m2=omr[0],
                      e2=omr[1])
    out = torch.empty(rows, C, dtype=torch.int16, device=DEV)
    _lib.call("ivit_residual_requant_i16", _lib.ptr(dev(a)), a_bits, _lib.ptr(mp), _lib.ptr(ep), ma[0], ma[1],
              _lib.ptr(dev(res)), mr[0], mr[1], _lib.ptr(out), rows, C, H, H, ws, shift, st())
    assert np.array_equal(out.cpu().numpy().astype(np.int32), ref)
    # identity map
    out2 = torch.empty(rows, C, dtype=torch.int16, device=DEV)
    _lib.call("ivit_residual_requant_i16", _lib.ptr(dev(a)), a_bits, _lib.ptr(mp), _lib.ptr(ep), ma[0], ma[1],
              _lib.ptr(dev(res)), mr[0], mr[1], _lib.ptr(out2), rows, C, 0, 0, 0, 0, st())
    ref2 = orc.requant(np.ascontiguousarray(k), oma[0], oma[1], 16, z2=res.astype(np.int32), m2=omr[0], e2=omr[1])
    assert np.array_equal(out2.cpu().numpy().astype(np.int32), ref2)


def test_residual_requant_i16_rejects_bad_arguments():
    a = dev(np.zeros((49, 8), np.int8))
    r = dev(np.zeros((49, 8), np.int16))
    o = torch.empty(49, 8, dtype=torch.int16, device=DEV)
    with pytest.raises(_lib.IvitError):   # H not a multiple of the window
        _lib.call("ivit_residual_requant_i16", _lib.ptr(a), 8, None, None, 1 << 30, 30, _lib.ptr(r), 1 << 30, 30,
                  _lib.ptr(o), 49, 8, 7, 7, 4, 0, st())
    with pytest.raises(_lib.IvitError):   # int32 accumulators without their requantiser table
        _lib.call("ivit_residual_requant_i16", _lib.ptr(a), 32, None, None, 1 << 30, 30, _lib.ptr(r), 1 << 30, 30,
                  _lib.ptr(o), 49, 8, 0, 0, 0, 0, st())


@pytest.mark.parametrize("rows,C,amp", [(300, 96, 30000), (64, 384, 3000), (50, 768, 32767), (33, 1536, 200), (7, 100, 5)])
def test_layernorm_i16_i8(rows, C, amp):
    rng = np.random.default_rng(C + amp)
    x = rng.integers(-amp, amp + 1, size=(rows, C)).astype(np.int16)
    x[0] = 0  # constant row: var = 0
    x[1] = amp
    gamma = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.1).astype(np.float32)
    s_out = np.float32(4.0 / 127.0)
    lp = LayerNormParams(gamma, beta, s_out)
    y, s_ln, _ = orc.layernorm(x.astype(np.int32), gamma, beta)
    z = orc.roundtrip(y, s_ln)
    om, oe = orc.dyadic(s_ln, s_out)
    ref = orc.requant(z, om, oe, 8)
    out = torch.zeros(rows, C + 16, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8", _lib.ptr(dev(x)), rows, C, _lib.ptr(dev(lp.bias_int)), _lib.ptr(dev(lp.s_ln)),
              _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out), C + 16, 0, 0, 0, 0, st())
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :C].astype(np.int32), ref)
    assert not got[:, C:].any()


@pytest.mark.parametrize("M,N,K", [(3136, 96, 384), (6272, 192, 768), (1000, 384, 1536), (2600, 768, 3072)])
def test_gemm_requant_residual_i16(M, N, K):
    """mlp.fc2 + mlp.qact2 + 16-bit residual QuantAct in one kernel == the two-kernel form and the oracle"""
    rng = np.random.default_rng(M + N)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    res = rng.integers(-32768, 32768, size=(M, N)).astype(np.int16)
    m, e = orc.dyadic(rng.uniform(2e-5, 6e-5, size=N).astype(np.float32), np.float32(1.0))
    k8 = orc.requant(orc.gemm_i8(A, W, b), m, e, 8)
    dA, dW, db, dres = dev(A), dev(W), dev(b), dev(res)
    dm, de = dev(m.astype(np.uint32).view(np.int32)), dev(e.astype(np.int32))
    for s_main, s_res, s_out in [(0.031, 0.0007, 0.0009), (0.5, 0.25, 0.5), (0.02, 0.003, 0.0001)]:
        m1, e1 = sme(s_main, s_out)
        m2, e2 = sme(s_res, s_out)
        exp = np.clip(np.rint(k8.astype(np.float64) * m1 / 2.0 ** e1) + np.rint(res.astype(np.float64) * m2 / 2.0 ** e2),
                      -32768, 32767).astype(np.int16)
        out = torch.zeros(M, N, dtype=torch.int16, device=DEV)
        _lib.call("ivit_gemm_i8_requant_residual_i16", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(dm), _lib.ptr(de),
                  _lib.ptr(dres), N, m1, e1, m2, e2, _lib.ptr(out), N, M, N, K, st())
        assert np.array_equal(out.cpu().numpy(), exp), (s_main, s_res, s_out)
        # the two-kernel form it replaces
        k8d = torch.empty(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(dm), _lib.ptr(de),
                  _lib.ptr(k8d), N, M, N, K, st())
        out2 = torch.zeros(M, N, dtype=torch.int16, device=DEV)
        _lib.call("ivit_residual_requant_i16", _lib.ptr(k8d), 8, None, None, m1, e1, _lib.ptr(dres), m2, e2, _lib.ptr(out2), M, N,
                  0, 0, 0, 0, st())
        assert torch.equal(out, out2)
        # the weights-in-registers kernel (fragment-packed W): what the Swin engine uses where K % 192 == 0 and M >= 2048
        if M >= 2048 and N >= 128 and N % 64 == 0 and K % 192 == 0:
            Wf = torch.zeros((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
            _lib.call("ivit_pack_weight_frags_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
            out3 = torch.zeros(M, N, dtype=torch.int16, device=DEV)
            _lib.call("ivit_gemm_i8_requant_residual_i16_ex", _lib.ptr(dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(dm), _lib.ptr(de),
                      _lib.ptr(dres), N, m1, e1, m2, e2, _lib.ptr(out3), N, M, N, K, 8, st())
            assert torch.equal(out, out3)


@pytest.mark.parametrize("regime", ["tiny_gamma", "big_bias", "saturating", "vanishing"])
@pytest.mark.parametrize("C", [96, 384])
def test_layernorm_i16_certificate_regimes(regime, C):
    """float32 bracket certificate of the tiled int16 LayerNorm (and its literal fallback) at unusual magnitudes"""
    rows = 4000
    rng = np.random.default_rng(len(regime) * 100 + C)
    x = np.clip(np.rint(rng.normal(0, rng.uniform(5, 6000, size=(rows, 1)), size=(rows, C))), -32768, 32767).astype(np.int16)
    gamma = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.1).astype(np.float32)
    scale_out = 0.8
    if regime == "tiny_gamma":
        gamma = rng.uniform(1e-4, 3e-3, size=C).astype(np.float32)
        beta = (gamma * rng.normal(0, 0.2, size=C)).astype(np.float32)
    elif regime == "big_bias":
        beta = rng.normal(0, 8.0, size=C).astype(np.float32)
    elif regime == "saturating":
        scale_out = 0.02
    elif regime == "vanishing":
        scale_out = 300.0
    y, s_ln, _ = orc.layernorm(x.astype(np.int32), gamma, beta)
    s_out = np.float32(2.0 ** np.ceil(np.log2(max(float(np.abs(y * s_ln).max()), 1e-30) / 127 * scale_out)))
    lp = LayerNormParams(gamma, beta, s_out)
    om, oe = orc.dyadic(s_ln, s_out)
    ref = orc.requant(orc.roundtrip(y, s_ln), om, oe, 8)
    out = torch.zeros(rows, C, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8", _lib.ptr(dev(x)), rows, C, _lib.ptr(dev(lp.bias_int)), _lib.ptr(dev(lp.s_ln)),
              _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out), C, 0, 0, 0, 0, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, ref), f"{regime}: {(got != ref).sum()} of {got.size} differ"


def test_swin_aliased_workspaces_equal_separate_buffers():
    """IntSwinEngine._compact changes no bit"""
    from ivit_amd.checkpoint import load_synthetic_model
    fs, ranges, cfg, meta, z = load_synthetic_model("swin_tiny")
    eng = IntSwinEngine(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"], device=DEV, max_batch=4)
    imgs = torch.from_numpy(synth.make_images(4, 31)).to(DEV)
    li_a = eng.forward(imgs)[0].cpu().numpy().copy()
    eng._compact(False)
    li_o = eng.forward(imgs)[0].cpu().numpy().copy()
    eng._compact(True)
    assert np.array_equal(li_a, li_o)


def test_layernorm_i16_i8_window_order():
    rng = np.random.default_rng(5)
    B, H, ws, shift, C = 2, 14, 7, 3, 96
    rows = B * H * H
    x = rng.integers(-20000, 20000, size=(rows, C)).astype(np.int16)
    gamma = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.1).astype(np.float32)
    lp = LayerNormParams(gamma, beta, np.float32(4.0 / 127.0))
    args = (_lib.ptr(dev(lp.bias_int)), _lib.ptr(dev(lp.s_ln)), _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)))
    plain = torch.empty(rows, C, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8", _lib.ptr(dev(x)), rows, C, *args, _lib.ptr(plain), C, 0, 0, 0, 0, st())
    wino = torch.empty(rows, C, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8", _lib.ptr(dev(x)), rows, C, *args, _lib.ptr(wino), C, H, H, ws, shift, st())
    p = plain.cpu().numpy().reshape(B, H, H, C)
    exp = orc._win_partition(np.roll(p, (-shift, -shift), axis=(1, 2)), ws).reshape(rows, C)
    assert np.array_equal(wino.cpu().numpy(), exp)


def test_patch_merge_i16():
    rng = np.random.default_rng(9)
    B, H, W, C = 3, 8, 6, 20
    x = rng.integers(-32768, 32768, size=(B, H, W, C)).astype(np.int16)
    ref = np.concatenate([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    out = torch.empty(B * H * W * C, dtype=torch.int16, device=DEV)
    _lib.call("ivit_patch_merge_i16", _lib.ptr(dev(x)), _lib.ptr(out), B, H, W, C, st())
    assert np.array_equal(out.cpu().numpy().reshape(ref.shape), ref)


def test_avgpool_requant_i8():
    rng = np.random.default_rng(11)
    B, T, C = 5, 49, 768
    x = rng.integers(-128, 128, size=(B, T, C)).astype(np.int8)
    m, e = sme(0.031, 0.027)
    om, oe = ome(0.031, 0.027)
    mean_f = (x.astype(np.int32).sum(axis=1).astype(np.float32) / np.float32(T)).astype(np.float32)
    ref = orc.requant(np.rint(mean_f).astype(np.float32), om, oe, 8)
    out = torch.empty(B, C, dtype=torch.int8, device=DEV)
    _lib.call("ivit_avgpool_requant_i8", _lib.ptr(dev(x)), _lib.ptr(out), B, T, C, m, e, st())
    assert np.array_equal(out.cpu().numpy().astype(np.int32), ref)


def test_patchify_ld():
    rng = np.random.default_rng(13)
    B, hw, patch = 2, 32, 4
    img = rng.standard_normal((B, 3, hw, hw)).astype(np.float32)
    s0 = np.float32(4.0 / 127.0)
    inv = float(np.float32(1.0) / s0)
    g = hw // patch
    A = torch.full((B * g * g, 64), 77, dtype=torch.int8, device=DEV)
    _lib.call("ivit_quantize_patchify_ld_f32_i8", _lib.ptr(dev(img)), _lib.ptr(A), 64, B, 3, hw, patch, inv, st())
    k0 = orc.quant_sym(img, s0, 8)
    ref = k0.reshape(B, 3, g, patch, g, patch).transpose(0, 2, 4, 1, 3, 5).reshape(B * g * g, 48)
    got = A.cpu().numpy()
    assert np.array_equal(got[:, :48].astype(np.int32), ref)
    assert (got[:, 48:] == 77).all()


# ----------------------------------------------------------------------------------- window attention
def window_attention_ref(q8, k8, v8, bias_add, mask_add, nW, me_s, me_b, s_attn, me_o):
    """oracle composition of swin_quant.py:137-161 on integers; q8/k8/v8 [B_, nH, N, hd]"""
    B_, nH, N, hd = q8.shape
    S = np.einsum("bhqd,bhkd->bhqk", q8.astype(np.int64), k8.astype(np.int64)).astype(np.int32)
    kS = orc.requant(S.reshape(-1, N), me_s[0], me_s[1], 8).reshape(B_, nH, N, N)
    lin = orc.requant(kS.reshape(-1, N), me_b[0], me_b[1], 32).reshape(B_, nH, N, N)   # RNE(kS * Mb), no clamp yet
    kA = np.clip(lin + bias_add[None].astype(np.int32), -128, 127)
    if mask_add is not None:
        kA = (kA.reshape(B_ // nW, nW, nH, N, N) + mask_add[None, :, None].astype(np.int32)).reshape(B_, nH, N, N)
    Pm = orc.shiftmax(kA, s_attn)
    O = np.einsum("bhqk,bhkd->bqhd", Pm.astype(np.int64), v8.astype(np.int64)).astype(np.int32)
    return orc.requant(O.reshape(-1, hd), me_o[0], me_o[1], 8).reshape(B_, N, nH * hd), kA, Pm


@pytest.mark.parametrize("B_,nW,nH,N,s_attn,masked", [(8, 4, 3, 49, 0.25, True), (6, 1, 6, 49, 0.5, False),
                                                      (4, 4, 2, 16, 1.0, True), (3, 1, 24, 49, 2.0, False),
                                                      (5, 1, 1, 64, 0.125, False), (2, 2, 2, 9, 0.0625, True)])
@pytest.mark.parametrize("pow2_scores", [False, True])
def test_window_attention(B_, nW, nH, N, s_attn, masked, pow2_scores):
    """pow2_scores: both score multipliers powers of two, as in the power-of-two regime of the engine: the kernel's float32 form of the
    two requantisations (round 4), compared with the oracle AND with its float64 form (lab bit 23); else the float64 form"""
    rng = np.random.default_rng(B_ * 1000 + N)
    hd = 32
    qkv = rng.integers(-128, 128, size=(3, B_, nH, N, hd)).astype(np.int8)
    qkv[0, 0, 0, 0] = 127   # saturating scores
    qkv[1, 0, 0] = 127
    s_S = np.float32(2.0 ** -9 * (1.0 if pow2_scores else 0.9))
    s_at = np.float32(s_attn)
    ms, omS = sme(s_S, s_at), ome(s_S, s_at)
    mb, omB = sme(s_at, s_at * np.float32(1.0)), ome(s_at, s_at)
    if N == 49:  # a non-trivial qact_attn1 -> qact2 ratio too (0.5: every odd kS is an exact tie of the second requantisation)
        ratio = np.float32(0.5 if pow2_scores else 0.75)
        mb, omB = sme(s_at * ratio, s_at), ome(s_at * ratio, s_at)
    if pow2_scores:
        assert int(ms[0]) & (int(ms[0]) - 1) == 0 and int(mb[0]) & (int(mb[0]) - 1) == 0
    mo, omO = sme(np.float32(2.0 ** -7 * 0.05), 0.043), ome(np.float32(2.0 ** -7 * 0.05), 0.043)
    bias_add = rng.integers(-60, 61, size=(nH, N, N)).astype(np.int16)
    bias_pad = np.full((nH, N, 64), 99, np.int16)      # pad entries must never be used
    bias_pad[:, :, :N] = bias_add
    mask_add, region, mval = None, None, 0
    if masked:
        mval = int(np.float32(-100.0) / s_at)
        region = np.full((nW, 64), 200, np.uint8)
        region[:, :N] = rng.integers(0, 4, size=(nW, N))
        mask_add = np.where(region[:, :N, None] != region[:, None, :N], mval, 0).astype(np.int16)
    ref, kA, Pm = window_attention_ref(qkv[0], qkv[1], qkv[2], bias_add, mask_add, nW, omS, omB, s_at, omO)
    ld = nH * hd + 32
    out = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
    _lib.call("ivit_window_attention_i8", _lib.ptr(dev(qkv)), _lib.ptr(out), ld, _lib.ptr(dev(bias_pad)),
              _lib.ptr(None if region is None else dev(region)), mval, B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1],
              float(s_at), mo[0], mo[1], st())
    got = out.cpu().numpy()
    assert np.array_equal(got[:, : nH * hd].astype(np.int32).reshape(B_, N, nH * hd), ref)
    assert not got[:, nH * hd:].any()
    assert Pm.max() > 0
    if pow2_scores:
        out64 = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
        with _lib.lab_session():
            _lib.call("ivit_debug_ln_ablate", 1 << 23)
            _lib.call("ivit_window_attention_i8", _lib.ptr(dev(qkv)), _lib.ptr(out64), ld, _lib.ptr(dev(bias_pad)),
                      _lib.ptr(None if region is None else dev(region)), mval, B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1],
                      float(s_at), mo[0], mo[1], st())
        assert np.array_equal(out64.cpu().numpy(), got)
    # the same rows at their image positions (window reverse + roll back in the store address)
    ws_ = int(round(np.sqrt(N)))
    if ws_ * ws_ == N and B_ % nW == 0:
        from ivit_amd.swin_engine import window_row_map
        gh, gw = {1: (1, 1), 2: (1, 2), 4: (2, 2)}[nW]
        H, W = gh * ws_, gw * ws_
        for shift in sorted({0, ws_ // 2}):
            out2 = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
            _lib.call("ivit_window_attention_i8_unwindow", _lib.ptr(dev(qkv)), _lib.ptr(out2), ld, _lib.ptr(dev(bias_pad)),
                      _lib.ptr(None if region is None else dev(region)), mval, B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1],
                      float(s_at), mo[0], mo[1], None, None, H, W, ws_, shift, st())
            dst = window_row_map(B_ // nW, H, W, ws_, shift)        # image row r -> its row in window order
            assert np.array_equal(out2.cpu().numpy(), got[dst]), shift
        with pytest.raises(_lib.IvitError, match="do not describe"):
            _lib.call("ivit_window_attention_i8_unwindow", _lib.ptr(dev(qkv)), _lib.ptr(out), ld, _lib.ptr(dev(bias_pad)), None, 0, B_, nW,
                      nH, N, hd, ms[0], ms[1], mb[0], mb[1], float(s_at), mo[0], mo[1], None, None, H + 1, W, ws_, 0, st())


def test_window_attention_rejects_unsupported_geometry():
    q = dev(np.zeros(3 * 49 * 64, np.int8))
    o = torch.empty(49 * 64, dtype=torch.int8, device=DEV)
    b = dev(np.zeros(49 * 64, np.int16))
    with pytest.raises(_lib.IvitError):
        _lib.call("ivit_window_attention_i8", _lib.ptr(q), _lib.ptr(o), 64, _lib.ptr(b), None, 0, 1, 1, 1, 49, 64,
                  1 << 30, 40, 1 << 30, 31, 0.25, 1 << 30, 40, st())
    with pytest.raises(_lib.IvitError):
        _lib.call("ivit_window_attention_i8", _lib.ptr(q), _lib.ptr(o), 32, _lib.ptr(b), None, 0, 1, 1, 1, 65, 32,
                  1 << 30, 40, 1 << 30, 31, 0.25, 1 << 30, 40, st())


# ----------------------------------------------------------------------------------- whole model
def build_swin(max_batch, tag="swin_tiny"):
    z, meta, ranges = load_fixture(tag)
    cfg = synth.SWIN_CONFIGS[meta["factory"]]
    fs = synth.make_swin_float_state(meta["factory"], meta["weight_seed"])
    eng = IntSwinEngine(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"], device=DEV,
                        max_batch=max_batch)
    return eng, fs, ranges, cfg, meta, z


@pytest.mark.parametrize("tag", ["swin_tiny", "swin_small"])
def test_swin_golden_logits_and_taps(tag):
    eng, fs, ranges, cfg, meta, z = build_swin(4, tag)
    n = meta["n_images"]
    imgs = torch.from_numpy(synth.make_images(n, meta["image_seed"])).to(DEV)
    taps = {}
    li, lf, t1 = eng.forward(imgs, taps)
    torch.cuda.synchronize()
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    checked, bad = 0, []
    for name, t in taps.items():
        assert name in gold, name
        if crc(t.cpu().numpy().astype(np.int32)) != int(gold[name]):
            bad.append(name)
        checked += 1
    assert not bad, f"first differing taps: {bad[:6]}"
    nblk = sum(cfg["depths"])
    assert checked == 9 * nblk + 2 * (len(cfg["depths"]) - 1) + 5
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"])
    assert np.array_equal(lf.cpu().numpy().view(np.int32), z["logits_f32_bits"])
    assert np.array_equal(t1.cpu().numpy().astype(np.int64), z["top1"])


def test_swin_batch_invariance_and_oracle_on_fresh_images():
    eng, fs, ranges, cfg, meta, z = build_swin(16)
    imgs_np = synth.make_images(16, 5151)
    imgs = torch.from_numpy(imgs_np).to(DEV)
    li, lf, t1 = eng.forward(imgs)
    li = li.cpu().numpy().copy()
    om = orc.OracleSwin(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"])
    sub = [0, 15]
    ref = om.forward(imgs_np[sub])
    assert np.array_equal(li[sub], ref["logits_int32"])
    perm = np.random.default_rng(0).permutation(16)
    li2, _, _ = eng.forward(imgs[torch.from_numpy(perm).to(DEV)].contiguous())
    assert np.array_equal(li2.cpu().numpy(), li[perm])
    li3, _, _ = eng.forward(imgs[5:6].contiguous())
    assert np.array_equal(li3.cpu().numpy(), li[5:6])


def test_swin_graph_replay_matches_eager():
    eng, fs, ranges, cfg, meta, z = build_swin(4)
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    g1, _, _ = eng.forward_graph(imgs)
    assert np.array_equal(g1.cpu().numpy(), z["logits_int32"])
    g2, _, _ = eng.forward_graph(torch.flip(imgs, dims=[0]).contiguous())
    assert np.array_equal(g2.cpu().numpy(), z["logits_int32"][::-1])


def test_swin_base_widths_calibrated_on_gpu_match_oracle():
    """Swin-B channel widths (C = 128..1024, heads 4..32, two blocks per stage): ranges come from a calibration forward
    of the module mirror on the GPU (running-stat QuantActs, HIP min/max), snapped to powers of two; then the fused
    engine, the module-by-module path and the CPU oracle agree on the INT32 logits."""
    import ivit_amd as ivit
    import ivit_amd.quantization_utils as q
    from ivit_amd.swin_quant import SwinTransformer
    from functools import partial
    cfg = synth.SWIN_CONFIGS["swin_base_shallow"]
    fs = synth.make_swin_float_state("swin_base_shallow", 33)
    model = SwinTransformer(embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                            window_size=cfg["window"], norm_layer=partial(q.IntLayerNorm, eps=1e-6))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    assert not unexpected
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(2, 909)).to(DEV))           # calibration forward
    ranges = {}
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct) and name != "act_out":
            mx = max(-float(mod.x_min), float(mod.x_max))
            qmax = 2 ** (mod.activation_bit - 1) - 1
            p = int(np.ceil(np.log2(mx / qmax)))
            mod.x_max.fill_(qmax * 2.0 ** p)
            mod.x_min.fill_(-qmax * 2.0 ** p)
            ranges[name] = (np.float32(-qmax * 2.0 ** p), np.float32(qmax * 2.0 ** p))
    assert list(ranges) == synth.swin_qact_names(cfg["depths"])
    ivit.freeze_model(model)
    imgs_np = synth.make_images(2, 910)
    imgs = torch.from_numpy(imgs_np).to(DEV)
    with torch.no_grad():
        y_engine = model(imgs)                       # fused engine (head_dim 32 everywhere)
        model.use_engine = False
        y_modules = model(imgs)
    assert np.array_equal(y_engine.cpu().numpy().view(np.int32), y_modules.cpu().numpy().view(np.int32))
    om = orc.OracleSwin(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"])
    ref = om.forward(imgs_np)
    assert np.array_equal(y_engine.cpu().numpy().view(np.int32), ref["logits_f32"].view(np.int32))
    assert om.max_acc < 2 ** 24


def test_config5_batch_128_swin_tiny():
    """Config 5 at full size (batch 128: 401 408-row stage-0 GEMMs / LN16, 24 576 (window, head) pairs per attention
    launch): the reference's golden images embedded in the batch reproduce its INT32 logits, and the batch is permutation
    invariant at that size."""
    eng, fs, ranges, cfg, meta, z = build_swin(128)
    imgs_np = synth.make_images(128, 8128)
    gold = synth.make_images(meta["n_images"], meta["image_seed"])
    pos = [0, 63, 127][: meta["n_images"]]
    for p, g in zip(pos, gold):
        imgs_np[p] = g
    imgs = torch.from_numpy(imgs_np).to(DEV)
    li, lf, t1 = eng.forward(imgs)
    li = li.cpu().numpy().copy()
    t1 = t1.cpu().numpy().copy()
    assert np.array_equal(li[pos], z["logits_int32"])
    assert np.array_equal(t1[pos].astype(np.int64), z["top1"])
    perm = np.random.default_rng(5).permutation(128)
    li2, _, _ = eng.forward(imgs[torch.from_numpy(perm).to(DEV)].contiguous())
    assert np.array_equal(li2.cpu().numpy(), li[perm])
    # two fresh images of the batch against the CPU oracle
    om = orc.OracleSwin(fs, ranges, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["window"])
    sub = [17, 101]
    assert np.array_equal(li[sub], om.forward(imgs_np[sub])["logits_int32"])


@pytest.mark.parametrize("tag", ["swin_tiny_natural", "swin_small_natural"])
def test_swin_natural_scales_match_reference_end_to_end(tag):
    """Swin-T (and, since round 4, Swin-S: 18 blocks in stage 2) with its ranges AS CALIBRATED against the reference itself (fixture swin_tiny_natural.npz: all 192 QuantAct /
    Shiftmax / ShiftGELU taps, INT32 logits, top-1 of the reference's forward), module-by-module path AND fused engine.
    The reference's patch-embed LayerNorm takes its float32 mean over a TRANSPOSED view (layers_quant.py:198-201): ATen sums such
    a row with its outer-reduction cascade, not with the 32-partial-sum order of a contiguous row, and the 69 exact-tie rows of
    this fixture (1 in 96) are decided by that order (csrc/rowsum.h torch_outer_rowsum; the reference run with 1, 4 and 8
    threads gives identical taps at this batch: oracle/gen_golden.py)."""
    import zlib
    import ivit_amd as ivit
    import ivit_amd.quantization_utils as qu
    z, meta, ranges = load_fixture(tag)
    fs = synth.make_swin_float_state(meta["factory"], meta["weight_seed"])
    model = getattr(ivit, meta["factory"])()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, qu.QuantAct) and name in ranges:
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    order = synth.swin_qact_names(synth.SWIN_CONFIGS[meta["factory"]]["depths"])
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    # ---- module-by-module path
    model.use_engine = False
    got, seen = {}, []

    def hook(name):
        def fn(mod, inp, outp):
            y, s = outp
            got[name] = zlib.crc32(np.ascontiguousarray(torch.round(y / s).to(torch.int64).cpu().numpy().astype(np.int32)).tobytes())
            seen.append(name)
        return fn

    handles = [mod.register_forward_hook(hook(name)) for name, mod in model.named_modules()
               if isinstance(mod, (qu.QuantAct, qu.IVITIntSoftmax, qu.IVITIntGELU)) and name != "act_out"]
    with torch.no_grad():
        ym = model(imgs)
    for h in handles:
        h.remove()
    assert set(got) == set(gold), sorted(set(gold) ^ set(got))[:5]
    bad = [n for n in seen if got[n] != int(gold[n])]
    assert not bad, f"module path: {len(bad)} of {len(got)} taps differ from the reference, first {bad[:4]}"
    s_head = torch.from_numpy(z["head_scale"]).to(DEV)
    li_m = torch.round(ym / s_head).to(torch.int64).cpu().numpy().astype(np.int32)
    assert np.array_equal(li_m, z["logits_int32"])
    assert np.array_equal(ym.argmax(dim=1).cpu().numpy().astype(np.int64), z["top1"])
    # ---- fused engine
    model.use_engine = True
    assert model.engine_unsupported_reason() is None
    with torch.no_grad():
        model(imgs)
    eng = model._engine[2]
    assert eng.natural_sites > 30
    taps = {}
    li, lf, t1 = eng.forward(imgs, taps)
    torch.cuda.synchronize()
    bad = [n for n in order if n in taps and crc(taps[n].cpu().numpy().astype(np.int32)) != int(gold[n])]
    assert not bad, f"engine: {len(bad)} taps differ from the reference, first {bad[:4]}"
    assert len(taps) >= 119 if tag == "swin_tiny_natural" else len(taps) >= 200
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy().astype(np.int64), z["top1"])


# ----------------------------------------------------------------------------------- natural scales (Swin engine)
def test_layernorm_i16_natural_scale_kat(golden_dir):
    """the literal 16-bit LayerNorm kernel against the reference module at natural input scales (rows with exact mean ties)"""
    from ivit_amd.prepare import LayerNormParams, sym_scale
    ck = np.load(os.path.join(golden_dir, "compat_kat.npz"))
    for ci in ck["ln16_cases"]:
        c = f"ln16_{ci}_"
        q = ck[c + "q"]
        rows, Cn = q.shape
        lo, hi = ck[c + "range"]
        lp = LayerNormParams(ck[c + "gamma"], ck[c + "beta"], sym_scale(lo, hi, 8))
        from ivit_amd.prepare import markstein_division_ok
        assert markstein_division_ok(ck[c + "s"], 16)
        for fast in (0, 1):      # the literal one-wave-per-row kernel; the tiled kernel with the 3-instruction quotient
            out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
            _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(dev(q)), rows, Cn, float(ck[c + "s"]), fast, _lib.ptr(dev(lp.bias_int)),
                      _lib.ptr(dev(lp.s_ln)), _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out), Cn,
                      0, 0, 0, 0, st())
            assert np.array_equal(out.cpu().numpy().astype(np.int32), ck[c + "q_out"]), (ci, fast)


@pytest.mark.parametrize("B_,nW,nH,N,s_attn,masked", [(8, 4, 3, 49, 0.271, True), (6, 1, 6, 49, 0.1173, False),
                                                      (4, 4, 2, 16, 1.3, True), (3, 1, 4, 49, 0.25, True), (8, 4, 2, 49, 0.0613, True),
                                                      (4, 2, 3, 49, 0.3391, True)])
def test_window_attention_natural_scale(B_, nW, nH, N, s_attn, masked):
    """the literal Shiftmax inside the window-attention kernel (phi tables, float shift mask) vs the oracle composition; the
    last case has a power-of-two scale with a non-integer -100/s ... which is still an integer there: covers phi = identity"""
    rng = np.random.default_rng(B_ * 1000 + N + 7)
    hd = 32
    qkv = rng.integers(-128, 128, size=(3, B_, nH, N, hd)).astype(np.int8)
    s_S = np.float32(2.0 ** -9 * 0.9)
    s_at = np.float32(s_attn)
    ms, omS = sme(s_S, s_at), ome(s_S, s_at)
    mb, omB = sme(s_at * np.float32(0.75), s_at), ome(s_at * np.float32(0.75), s_at)
    mo, omO = sme(np.float32(2.0 ** -7 * 0.05), 0.043), ome(np.float32(2.0 ** -7 * 0.05), 0.043)
    bias_add = rng.integers(-60, 61, size=(nH, N, N)).astype(np.int16)
    bias_pad = np.full((nH, N, 64), 99, np.int16)
    bias_pad[:, :, :N] = bias_add
    region, maskb = None, np.zeros((nW, N, N), bool)
    if masked:
        region = np.full((nW, 64), 200, np.uint8)
        region[:, :N] = rng.integers(0, 3, size=(nW, N))
        maskb = region[:, :N, None] != region[:, None, :N]
    # oracle composition: integers up to qact2, then Shiftmax on the float view x / s with the float mask
    S = np.einsum("bhqd,bhkd->bhqk", qkv[0].astype(np.int64), qkv[1].astype(np.int64)).astype(np.int32)
    kS = orc.requant(S.reshape(-1, N), omS[0], omS[1], 8).reshape(B_, nH, N, N)
    lin = orc.requant(kS.reshape(-1, N), omB[0], omB[1], 32).reshape(B_, nH, N, N)
    kA = np.clip(lin + bias_add[None].astype(np.int32), -128, 127).astype(np.float32)
    mfull = np.broadcast_to(maskb[None, :, None], (B_ // nW, nW, nH, N, N)).reshape(B_, nH, N, N)
    x = ((kA * s_at).astype(np.float32) + np.where(mfull, np.float32(-100.0), np.float32(0.0))).astype(np.float32)
    Pm = orc.shiftmax_xint((x / s_at).astype(np.float32), s_at)
    O = np.einsum("bhqk,bhkd->bqhd", Pm.astype(np.int64), qkv[2].astype(np.int64)).astype(np.int32)
    ref = orc.requant(O.reshape(-1, hd), omO[0], omO[1], 8).reshape(B_, N, nH * hd)
    qv = np.arange(-128, 128, dtype=np.float32)
    phi = ((qv * s_at).astype(np.float32) / s_at).astype(np.float32)
    phim = ((((qv * s_at).astype(np.float32) + np.float32(-100.0)).astype(np.float32)) / s_at).astype(np.float32)
    ld = nH * hd
    out = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
    _lib.call("ivit_window_attention_i8_compat", _lib.ptr(dev(qkv)), _lib.ptr(out), ld, _lib.ptr(dev(bias_pad)),
              _lib.ptr(None if region is None else dev(region)), -1, B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1],
              float(s_at), mo[0], mo[1], _lib.ptr(dev(phi)), _lib.ptr(dev(phim)), st())
    got = out.cpu().numpy().astype(np.int32).reshape(B_, N, nH * hd)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {got.size} differ"
    assert Pm.max() > 0
    # round 4: the table form (exp_int of every (row max, score) pair from the host) where the host can prove it equal
    from ivit_amd.prepare import window_shiftexp_band
    band, bw = window_shiftexp_band(s_at, masked)
    assert (band is None) == (s_attn == 1.3), "s = 1.3: -100 / s = -77 does not push masked scores below every unmasked one"
    if band is not None:
        assert bw % 16 == 0 and np.all(band[:, bw - 1] == band[0, bw - 1]) and band.shape[0] in (1, 256)
        # 0.271 and 0.25: exp_int depends on the distance to the maximum alone (one row); 0.1173, 0.0613, 0.3391: 256 rows
        assert (band.shape[0] == 1) == (s_attn in (0.271, 0.25)), band.shape
        out_b = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
        _lib.call("ivit_window_attention_i8_band", _lib.ptr(dev(qkv)), _lib.ptr(out_b), ld, _lib.ptr(dev(bias_pad)),
                  _lib.ptr(None if region is None else dev(region)), B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1], float(s_at),
                  mo[0], mo[1], _lib.ptr(dev(band)), bw, band.shape[0], 0, 0, 0, 0, st())
        gb = out_b.cpu().numpy().astype(np.int32).reshape(B_, N, nH * hd)
        assert np.array_equal(gb, ref), f"band form: {(gb != ref).sum()} of {gb.size} differ"
        if band.shape[0] == 1:      # the same values as 256 identical rows: the per-maximum form of the kernel
            out_r = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
            _lib.call("ivit_window_attention_i8_band", _lib.ptr(dev(qkv)), _lib.ptr(out_r), ld, _lib.ptr(dev(bias_pad)),
                      _lib.ptr(None if region is None else dev(region)), B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1], float(s_at),
                      mo[0], mo[1], _lib.ptr(dev(np.repeat(band, 256, axis=0))), bw, 256, 0, 0, 0, 0, st())
            assert np.array_equal(out_r.cpu().numpy(), out_b.cpu().numpy())
        ws_ = int(round(np.sqrt(N)))
        if ws_ * ws_ == N and B_ % nW == 0:      # image-ordered rows, as the engine's fused projection wants them
            from ivit_amd.swin_engine import window_row_map
            gh, gw = {1: (1, 1), 2: (1, 2), 4: (2, 2)}[nW]
            H, W = gh * ws_, gw * ws_
            out2 = torch.zeros(B_ * N, ld, dtype=torch.int8, device=DEV)
            _lib.call("ivit_window_attention_i8_band", _lib.ptr(dev(qkv)), _lib.ptr(out2), ld, _lib.ptr(dev(bias_pad)),
                      _lib.ptr(None if region is None else dev(region)), B_, nW, nH, N, hd, ms[0], ms[1], mb[0], mb[1], float(s_at),
                      mo[0], mo[1], _lib.ptr(dev(band)), bw, band.shape[0], H, W, ws_, ws_ // 2, st())
            dst = window_row_map(B_ // nW, H, W, ws_, ws_ // 2)
            assert np.array_equal(out2.cpu().numpy(), out_b.cpu().numpy()[dst])
        with pytest.raises(_lib.IvitError, match="band table"):
            _lib.call("ivit_window_attention_i8_band", _lib.ptr(dev(qkv)), _lib.ptr(out_b), ld, _lib.ptr(dev(bias_pad)), None, B_, nW, nH, N,
                      hd, ms[0], ms[1], mb[0], mb[1], float(s_at), mo[0], mo[1], _lib.ptr(dev(band)), bw + 3, band.shape[0], 0, 0, 0, 0, st())


def test_swin_natural_scales_engine_equals_module_path():
    """Swin-T with ranges as calibrated: the engine (literal 16-bit LayerNorm, literal Shiftmax on phi tables, remapped
    ShiftGELU table) and the module-by-module path (literal float kernels) are two implementations of the same arithmetic:
    identical float logits and identical integer taps"""
    import ivit_amd as ivit
    import ivit_amd.quantization_utils as qu
    z, meta, ranges = load_fixture("swin_tiny_natural")
    fs = synth.make_swin_float_state(meta["factory"], meta["weight_seed"])
    model = getattr(ivit, meta["factory"])()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, qu.QuantAct) and name in ranges:
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    assert model.engine_unsupported_reason() is None
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    with torch.no_grad():
        ye = model(imgs)
        eng = model._engine[2]
        assert eng.natural_sites > 30
        taps = {}
        eng.forward(imgs, taps)
        model.use_engine = False
        got = {}

        def hook(name):
            def fn(mod, inp, outp):
                y, s = outp
                got[name] = torch.round(y / s).to(torch.int32)
            return fn

        for name, mod in model.named_modules():
            if isinstance(mod, qu.QuantAct) and name != "act_out":
                mod.register_forward_hook(hook(name))
        ym = model(imgs)
    # first the taps in forward order (a difference is reported where it starts), then the logits
    for name in synth.swin_qact_names(synth.SWIN_CONFIGS[meta["factory"]]["depths"]):
        if name in taps and name in got:
            a, b = taps[name].cpu().numpy().astype(np.int32).reshape(-1), got[name].cpu().numpy().reshape(-1)
            assert a.size == b.size and np.array_equal(a, b), f"tap {name}: {(a != b).sum()} of {a.size} differ"
    assert np.array_equal(ye.cpu().numpy().view(np.int32), ym.cpu().numpy().view(np.int32))


@pytest.mark.parametrize("rows,Cn,s", [(5000, 96, 0.000913), (3001, 192, 0.0004471), (999, 384, 0.00171), (777, 768, 0.000613),
                                       (130, 1536, 0.0009), (64, 3072, 0.0011), (4099, 128, 0.00077), (2050, 256, 0.0021)])
@pytest.mark.parametrize("sum_form", ["product", "lab_sums_through_lds"])
def test_layernorm_i16_natural_scale_random_vs_oracle(rows, Cn, s, sum_form):
    """every LPR / NJ instantiation of the tiled natural-scale kernel (and the literal kernel for C > 1536) against the oracle,
    whose restatement is pinned by the reference KATs; a third of the rows are exact ties of the mean.  Round 4: the float32 row
    sums in torch's order stay in registers (DPP) for C < 512; the lab build keeps the round-3 form (through LDS) for A/B"""
    if sum_form != "product":
        with _lib.lab_session():
            _lib.call("ivit_debug_ln_ablate", 1 << 20)
            _ln16_natural_random(rows, Cn, s)
        return
    _ln16_natural_random(rows, Cn, s)


def _ln16_natural_random(rows, Cn, s):
    from ivit_amd.prepare import LayerNormParams, markstein_division_ok
    rng = np.random.default_rng(rows + Cn)
    s = np.float32(s)
    q = np.clip(np.rint(rng.normal(rng.normal(0, 2000, size=(rows, 1)), rng.uniform(500, 8000, size=(rows, 1)), size=(rows, Cn))),
                -32768, 32767).astype(np.int32)
    for r in range(0, rows, 3):
        d = Cn // 2 + Cn * int(rng.integers(-100, 100)) - int(q[r].sum())
        for c in rng.permutation(Cn):
            if d == 0:
                break
            nv = int(np.clip(q[r, c] + d, -32768, 32767))
            d -= nv - q[r, c]
            q[r, c] = nv
    gamma = rng.uniform(0.5, 1.5, size=Cn).astype(np.float32)
    beta = rng.normal(0, 0.1, size=Cn).astype(np.float32)
    y, s_ln, _ = orc.layernorm_scaled(q, s, gamma, beta)
    s_out = np.float32(np.abs(y * s_ln).max() / 127 * 0.83)
    m, e = orc.dyadic(s_ln, s_out)
    exp = orc.requant(orc.roundtrip(y, s_ln), m, e, 8)
    lp = LayerNormParams(gamma, beta, s_out)
    assert markstein_division_ok(s, 16)
    out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(dev(q.astype(np.int16))), rows, Cn, float(s), 1, _lib.ptr(dev(lp.bias_int)),
              _lib.ptr(dev(lp.s_ln)), _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out), Cn, 0, 0, 0, 0, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"


@pytest.mark.parametrize("rows,Cn,outer,s", [(6272, 96, 3136, 0.000913), (4096, 64, 64, 0.00171), (2048, 128, 1024, 0.0004471),
                                             (1280, 192, 320, 0.00077), (3136, 96, 49, 0.000913)])
def test_layernorm_i16_natural_outer_order_register_form_equals_lds_form(rows, Cn, outer, s):
    """Swin stage 0: the mean over the patch embedding's transposed view (IVIT_LN_OUTER_MEAN: torch's outer-reduction order).  Round 4
    keeps that sum in registers (groups of 16 in sequence across a lane pair, DPP) when the extent has no tail columns (outer % 32
    == 0) and C < 256; the round-3 form (a lane sums a row serially from LDS: pinned by the reference's end-to-end goldens, 69
    exact-tie rows) stays in the lab build and for outer = 49.  Half of the rows here are exact ties of the mean."""
    from ivit_amd.prepare import LayerNormParams, markstein_division_ok
    rng = np.random.default_rng(rows + Cn + outer)
    s = np.float32(s)
    assert markstein_division_ok(s, 16)
    q = np.clip(np.rint(rng.normal(rng.normal(0, 2000, size=(rows, 1)), rng.uniform(500, 8000, size=(rows, 1)), size=(rows, Cn))),
                -32768, 32767).astype(np.int32)
    for r in range(0, rows, 2):
        d = Cn // 2 + Cn * int(rng.integers(-100, 100)) - int(q[r].sum())
        for c in rng.permutation(Cn):
            if d == 0:
                break
            nv = int(np.clip(q[r, c] + d, -32768, 32767))
            d -= nv - q[r, c]
            q[r, c] = nv
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=Cn).astype(np.float32), rng.normal(0, 0.1, size=Cn).astype(np.float32), np.float32(0.031))
    qd = dev(q.astype(np.int16))

    def run():
        out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
        _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(qd), rows, Cn, float(s), 1 | (outer << 8), _lib.ptr(dev(lp.bias_int)),
                  _lib.ptr(dev(lp.s_ln)), _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out), Cn, 0, 0, 0, 0, st())
        return out.cpu().numpy()

    got = run()
    with _lib.lab_session():
        _lib.call("ivit_debug_ln_ablate", 1 << 20)
        ref = run()
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {got.size} differ"
    # and the order matters on this data: the contiguous-row order gives other bytes on some tie rows
    out_c = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(qd), rows, Cn, float(s), 1, _lib.ptr(dev(lp.bias_int)), _lib.ptr(dev(lp.s_ln)),
              _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(out_c), Cn, 0, 0, 0, 0, st())
    if Cn >= 96:
        assert (out_c.cpu().numpy() != got).any()


def test_swin_uint8_input_equals_the_float_pipeline():
    """uint8 pixels (3 x 256 table of ToTensor + Normalize + the input QuantAct, 4 x 4 patches with the K padding) == the float32
    images the data pipeline would hand over: INT32 logits"""
    eng = build_swin(3)[0]
    rng = np.random.default_rng(78)
    u8 = torch.from_numpy(rng.integers(0, 256, size=(3, 3, 224, 224), dtype=np.uint8)).to(DEV)
    from ivit_amd.prepare import IMAGENET_MEAN, IMAGENET_STD
    mean = torch.tensor(IMAGENET_MEAN, device=DEV).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=DEV).view(1, 3, 1, 1)
    xf = ((u8.float().div(255) - mean) / std).contiguous()
    li_f = eng.forward(xf)[0].clone()
    li_u = eng.forward(u8)[0].clone()
    assert torch.equal(li_u, li_f) and li_f.abs().max() > 0
