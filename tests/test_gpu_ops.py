"""GPU parity, operator level: every HIP kernel (through the C ABI) against the CPU oracle and the
reference-generated known-answer vectors.  Bit-exact: these are integer results."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ivit = pytest.importorskip("ivit_amd")
from ivit_amd import _lib  # noqa: E402
from ivit_amd.prepare import dyadic  # noqa: E402

DEV = "cuda:0"


_KEEP = []  # device tensors whose raw pointers were handed to the C ABI stay alive for the whole test


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


def rand_me(rng, n, lo_exp=-12, hi_exp=-6):
    """random requant ratios -> (m uint32, e int32) via the product's own dyadic()"""
    pre = (rng.uniform(0.5, 1.0, size=n) * 2.0 ** rng.integers(lo_exp, hi_exp, size=n)).astype(np.float32)
    m, e = dyadic(pre, np.float32(1.0))
    return m, e


def me_dev(m, e):
    return dev(m.view(np.int32)), dev(e)


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_kat.npz"))


# ----------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 192, 192), (197 * 3, 768, 768), (1000, 384, 1536),
                                   (64, 2304, 768), (129, 208, 128),
                                   # M >= 2048 and N >= 128: the 256x128-tile LDS-DMA kernel (1, 2, 3 and many K steps,
                                   # ragged token and channel tails)
                                   (2048, 128, 64), (2304, 256, 128), (2500, 208, 192), (197 * 16, 768, 768),
                                   (2100, 384, 3072),
                                   # N % 256 == 0 and K % 128 == 0: the 256x256x128 kernel (1, 2, 3, many K steps)
                                   (2048, 256, 128), (2300, 512, 256), (197 * 12, 768, 384), (2100, 768, 3072),
                                   (197 * 11, 2304, 768)])
def test_gemm_requant(M, N, K):
    rng = np.random.default_rng(M * 7 + N + K)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    acc = orc.gemm_i8(A, W, b)
    exp = orc.requant(acc, m.astype(np.float64), e, 8)
    out = torch.empty(M, N, dtype=torch.int8, device=DEV)
    md, ed = me_dev(m, e)
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, _lib.ptr(dev(b)), _lib.ptr(md),
              _lib.ptr(ed), _lib.ptr(out), N, M, N, K, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"
    assert np.abs(exp).max() > 10


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (2304, 256, 128), (2304, 128, 192)])
def test_gemm_requant_ties_and_exact_fallback(M, N, K):
    """(a) multipliers that are exact float32 values (1/16, 3/32): one accumulator in 16 is an exact tie and must
    round half to even on the float32 fast path; (b) accumulators beyond 2^22 (huge bias) and multipliers with 31
    significant bits: the float64 fallback of the epilogue."""
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-20, 21, size=(N, K)).astype(np.int8)
    b = rng.integers(-500, 500, size=N).astype(np.int32)
    m = np.full(N, 1 << 30, np.uint32)
    e = np.full(N, 34, np.int32)            # M = 2^30 / 2^34 = 1/16
    m[1::2] = 3 << 29                       # M = 3/32
    e[1::2] = 35
    b[5::16] = 30_000_000                   # |acc| > 2^22 -> exact path; tiny multiplier keeps it in range
    m[5::16] = (1 << 30) + 12345
    e[5::16] = 30 + 25
    b[6::16] = -2_000_000_000               # near the int32 limit
    m[6::16] = (1 << 31) - 1
    e[6::16] = 31 + 24
    acc = orc.gemm_i8(A, W, b)
    assert (np.abs(acc[:, 0::2] % 16) == 8).any(), "no exact ties generated"
    exp = orc.requant(acc, m.astype(np.float64), e, 8)
    out = torch.empty(M, N, dtype=torch.int8, device=DEV)
    md, ed = me_dev(m, e)
    _lib.call("ivit_gemm_i8_requant", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, _lib.ptr(dev(b)), _lib.ptr(md),
              _lib.ptr(ed), _lib.ptr(out), N, M, N, K, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"


def test_gemm_i32_and_bias_none():
    rng = np.random.default_rng(5)
    M, N, K = 70, 1000, 192
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    for bias in (b, None):
        out = torch.empty(M, N, dtype=torch.int32, device=DEV)
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K,
                  None if bias is None else _lib.ptr(dev(bias)), _lib.ptr(out), N, M, N, K, st())
        assert np.array_equal(out.cpu().numpy(), orc.gemm_i8(A, W, bias))


@pytest.mark.parametrize("M,N,K", [(70, 96, 128), (3000, 192, 192), (401, 384, 384), (2500, 768, 768)])
def test_gemm_requant_i16(M, N, K):
    """16-bit per-channel QuantAct from the accumulators (Swin attn.proj + attn.qact4), incl. saturation at +-32767/8"""
    rng = np.random.default_rng(M + N)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -8, 1)           # multipliers 2^-9 .. 1: results span and exceed the int16 range
    md, ed = me_dev(m, e)
    exp = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 16)
    for bias in (b, None):
        out = torch.zeros(M, N, dtype=torch.int16, device=DEV)
        _lib.call("ivit_gemm_i8_requant_i16", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, None if bias is None else _lib.ptr(dev(bias)),
                  _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), N, M, N, K, st())
        want = exp if bias is not None else orc.requant(orc.gemm_i8(A, W, None), m.astype(np.float64), e, 16)
        assert np.array_equal(out.cpu().numpy().astype(np.int32), want)
    assert (np.abs(exp) == 32767).any() or (exp == -32768).any()
    assert (np.abs(exp) < 30000).any()


@pytest.mark.parametrize("M,N,K", [(333, 96, 128), (1000, 200, 64)])
def test_gemm_requant_i16_residual_i16_small_shapes(M, N, K):
    """the same entry point on shapes only the 128 x 128-tile kernel takes (Swin attn.proj at C = 96; N not a multiple of 64)"""
    rng = np.random.default_rng(M + N)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    res = np.clip(np.rint(rng.normal(0, 9000, size=(M, N))), -32768, 32767).astype(np.int16)
    m, e = rand_me(rng, N, -8, 1)
    md, ed = me_dev(m, e)
    k16 = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 16)
    m1, e1 = dyadic(np.float32(0.3337), np.float32(0.0517))
    m2, e2 = dyadic(np.float32(0.0421), np.float32(0.0517))
    exp = orc.requant(k16, m1.astype(np.float64), e1, 16, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
    buf = dev(res).clone()
    _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, _lib.ptr(dev(b)), _lib.ptr(md), _lib.ptr(ed),
              _lib.ptr(buf), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(buf), N, M, N, K, 0, st())
    assert np.array_equal(buf.cpu().numpy().astype(np.int32), exp)


@pytest.mark.parametrize("M,N,K", [(2050, 768, 192), (4099, 384, 384), (128 * 37 + 64, 192, 768)])
def test_gemm_requant_i16_residual_i16(M, N, K):
    """ivit_gemm_i8_requant_i16_residual_i16_ex (projection / fc2 on a 16-bit residual stream: per-channel 16-bit QuantAct of the
    accumulators, then the two-operand 16-bit residual QuantAct) == oracle; ragged M (half tiles, partial last tile), saturation
    in both steps, scale pairs with exact ties, in place over the residual"""
    rng = np.random.default_rng(M + N)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    res = np.clip(np.rint(rng.normal(0, 9000, size=(M, N))), -32768, 32767).astype(np.int16)
    m, e = rand_me(rng, N, -8, 1)           # the 16-bit intermediate spans and exceeds the int16 range
    md, ed = me_dev(m, e)
    k16 = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 16)
    assert (np.abs(k16) >= 32767).any() and (np.abs(k16) < 30000).any()
    dA, dW, db, dres = dev(A), dev(W), dev(b), dev(res)
    Wf = torch.empty((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
    for s_main, s_res, s_out in [(0.7 * 2 ** -4, 2 ** -5, 2 ** -4), (2 ** -5, 2 ** -5, 2 ** -4), (1.0, 1.0, 1.0), (0.3337, 0.0421, 0.0517),
                                 (1.0, 1.0, 0.4)]:
        m1, e1 = dyadic(np.float32(s_main), np.float32(s_out))
        m2, e2 = dyadic(np.float32(s_res), np.float32(s_out))
        exp = orc.requant(k16, m1.astype(np.float64), e1, 16, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
        out = torch.full((M, N), 77, dtype=torch.int16, device=DEV)
        _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(dres), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(out), N, M, N, K, 8, st())
        got = out.cpu().numpy().astype(np.int32)
        assert np.array_equal(got, exp), (s_main, s_res, s_out, int((got != exp).sum()))
        inpl = dres.clone()
        _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(inpl), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(inpl), N, M, N, K, 8, st())
        assert torch.equal(inpl, out)
        # layouts = 0: row-major weights, the 128 x 128-tile kernel with the epilogue from its registers (any shape)
        out0 = torch.full((M, N), 55, dtype=torch.int16, device=DEV)
        _lib.call("ivit_gemm_i8_requant_i16_residual_i16_ex", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(dres), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(out0), N, M, N, K, 0, st())
        assert torch.equal(out0, out)


def test_gemm_mfma_layout_identity():
    """A = I-like selector with an asymmetric W catches any row/column swap of the MFMA maps."""
    K = 128
    A = np.zeros((128, K), np.int8)
    A[np.arange(128), np.arange(128) % K] = 1
    W = (np.arange(128)[:, None] * 3 + np.arange(K)[None, :] * 5) % 251 - 125
    W = W.astype(np.int8)
    out = torch.empty(128, 128, dtype=torch.int32, device=DEV)
    _lib.call("ivit_gemm_i8_i32", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, None, _lib.ptr(out), 128, 128, 128, K, st())
    assert np.array_equal(out.cpu().numpy(), orc.gemm_i8(A, W))


@pytest.mark.parametrize("M,N,K", [(333, 384, 384), (197 * 13, 768, 256), (2050, 192, 768), (2050, 768, 192)])
def test_gemm_requant_residual(M, N, K):
    rng = np.random.default_rng(11 + M)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    res = rng.integers(-128, 128, size=(M, N)).astype(np.int8)
    m, e = rand_me(rng, N, -16, -9)
    k3 = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    md, ed = me_dev(m, e)
    dA, dW, db, dres = dev(A), dev(W), dev(b), dev(res)
    # residual QuantAct scale pairs: the usual kind, ratios with exact ties (k * 0.5), identity, arbitrary floats
    pairs = [(0.7 * 2 ** -4, 2 ** -5, 2 ** -4), (2 ** -5, 2 ** -5, 2 ** -4), (1.0, 1.0, 1.0),
             (float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.05, 0.2))),
             (0.3337, 0.0421, 0.0517)]
    for s_main, s_res, s_out in pairs:
        m1, e1 = dyadic(np.float32(s_main), np.float32(s_out))
        m2, e2 = dyadic(np.float32(s_res), np.float32(s_out))
        exp = orc.requant(k3, m1.astype(np.float64), e1, 8, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
        out = torch.empty(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_residual", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db),
                  _lib.ptr(md), _lib.ptr(ed), _lib.ptr(dres), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]),
                  _lib.ptr(out), N, M, N, K, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), exp), (s_main, s_res, s_out)
        # in place: the output over the residual operand (how the engines keep one residual-stream buffer)
        inpl = dres.clone()
        _lib.call("ivit_gemm_i8_requant_residual", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db),
                  _lib.ptr(md), _lib.ptr(ed), _lib.ptr(inpl), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]),
                  _lib.ptr(inpl), N, M, N, K, st())
        assert torch.equal(inpl, out), (s_main, s_res, s_out)


@pytest.mark.parametrize("B,H", [(3, 3), (11, 6)])
def test_gemm_requant_qkv_layout(B, H):
    rng = np.random.default_rng(12 + B)
    T, hd = 197, 64
    Cn = H * hd
    M, N, K = B * T, 3 * Cn, Cn
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    exp = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    exp = exp.reshape(B, T, 3, H, hd).transpose(2, 0, 3, 1, 4)  # vit_quant.py:65-66
    out = torch.empty(3 * M * Cn, dtype=torch.int8, device=DEV)
    md, ed = me_dev(m, e)
    _lib.call("ivit_gemm_i8_requant_qkv", _lib.ptr(dev(A)), K, _lib.ptr(dev(W)), K, _lib.ptr(dev(b)), _lib.ptr(md),
              _lib.ptr(ed), _lib.ptr(out), T, H, hd, M, N, K, st())
    assert np.array_equal(out.cpu().numpy().astype(np.int32).reshape(3, B, H, T, hd), exp)


def test_gemm_both_kernels_agree():
    """same large problem through the LDS-DMA kernel and (forced) through the small-tile kernel"""
    rng = np.random.default_rng(77)
    M, N, K = 2600, 512, 384
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    md, ed = me_dev(m, e)
    dA, dW, db = dev(A), dev(W), dev(b)
    outs = []

    def run():
        out = torch.empty(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(out), N, M, N, K, st())
        outs.append(out.cpu().numpy())

    run()                                   # the product library (stateless: no knobs)
    with _lib.lab_session():                # libivit_hip_lab.so: the same sources with the kernel-form knobs
        for force in (0, 1, 2):
            _lib.call("ivit_debug_force_small_gemm", force)
            run()
        _lib.call("ivit_debug_force_small_gemm", 0)
        for flags in (1024, 2048, 4096):   # relaunch form; tail split; one workgroup per CU
            _lib.call("ivit_debug_set_gemm_flags", flags)
            run()
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    assert np.array_equal(outs[0].astype(np.int32), orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8))


def _block_layout_host(X):
    """numpy model of IVIT_LAYOUT_BLOCKS (include/ivit_hip.h): [rows, K] int8 -> flat bytes"""
    rows, K = X.shape
    R16 = (rows + 15) // 16
    out = np.zeros(R16 * 16 * K, dtype=np.int8)
    r = np.arange(rows)[:, None]
    c = np.arange(K)[None, :]
    rl = r & 15
    off = ((r >> 4) * (K >> 6) + (c >> 6)) * 1024 + (((rl << 2) + (((c >> 4) & 3) ^ ((rl >> 2) & 3))) << 4) + (c & 15)
    out[off.reshape(-1)] = X.reshape(-1)
    return out


@pytest.mark.parametrize("M,N,K", [(2600, 512, 384), (197 * 16, 768, 768), (2049, 128, 64)])
def test_gemm_block_layout_operands(M, N, K):
    """ivit_tile_operand_i8 == the documented layout; the GEMMs give identical results for every combination of
    row-major / block-layout operands (all three requantising epilogues); untile inverts tile"""
    rng = np.random.default_rng(M + K)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    md, ed = me_dev(m, e)
    dA, dW, db = dev(A), dev(W), dev(b)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    Wt = torch.zeros(N * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dA), K, M, K, _lib.ptr(At), st())
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wt), st())
    assert np.array_equal(At.cpu().numpy()[: _block_layout_host(A).size], _block_layout_host(A))
    assert np.array_equal(Wt.cpu().numpy(), _block_layout_host(W))
    back = torch.empty(M, K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_untile_operand_i8", _lib.ptr(At), M, K, _lib.ptr(back), K, st())
    assert np.array_equal(back.cpu().numpy(), A)
    exp = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    res = rng.integers(-128, 128, size=(M, N)).astype(np.int8)
    dres = dev(res)
    m1, e1 = dyadic(np.float32(0.7 * 2 ** -4), np.float32(2 ** -4))
    m2, e2 = dyadic(np.float32(2 ** -5), np.float32(2 ** -4))
    exp_res = orc.requant(exp, m1.astype(np.float64), e1, 8, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
    for lay in (0, 1, 2, 3):
        a_op = At if lay & 1 else dA
        w_op = Wt if lay & 2 else dW
        out = torch.empty(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(a_op), K, _lib.ptr(w_op), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(out), N, M, N, K, lay, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), exp), lay
        out = torch.empty(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(a_op), K, _lib.ptr(w_op), K, _lib.ptr(db), _lib.ptr(md),
                  _lib.ptr(ed), _lib.ptr(dres), N, int(m1[0]), int(e1[0]), int(m2[0]), int(e2[0]), _lib.ptr(out), N, M, N, K,
                  lay, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), exp_res), lay
    if M % 197 == 0 and N % 192 == 0:     # head-major q/k/v epilogue
        T, hd = 197, 64
        H = N // (3 * hd)
        outs = []
        for lay in (0, 3):
            out = torch.empty(3 * M * H * hd, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(At if lay else dA), K, _lib.ptr(Wt if lay else dW), K, _lib.ptr(db),
                      _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), T, H, hd, M, N, K, lay, st())
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
    # block-layout OUTPUT (the next GEMM's A operand written directly)
    R16 = (M + 15) // 16 * 16
    outb = torch.zeros(R16 * N, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wt), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
              _lib.ptr(outb), N, M, N, K, 7, st())
    if N % 64 == 0:
        assert np.array_equal(outb.cpu().numpy(), _block_layout_host(exp.astype(np.int8)))
    with pytest.raises(_lib.IvitError, match="persistent kernel"):   # small problems have no block-layout path
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(out), N, 512, N, K, 1, st())


def _frag_layout_host(W):
    """include/ivit_hip.h IVIT_W_FRAGS, by its documented formula"""
    N, K = W.shape
    n64 = (N + 63) // 64 * 64
    out = np.zeros(n64 * K, dtype=np.int8)
    n = np.arange(N)[:, None]
    k = np.arange(K)[None, :]
    off = ((n // 64) * (K // 64) + k // 64) * 4096 + (((n // 32) % 2 * 2 + (k // 32) % 2) * 2 + (k // 16) % 2) * 512 + (n % 32) * 16 + k % 16
    out[off.reshape(-1)] = W.reshape(-1)
    return out


def _frag16_layout_host(W):
    """include/ivit_hip.h IVIT_W_FRAGS16 (the v_mfma_i32_16x16x64_i8 fragment order), restated independently"""
    N, K = W.shape
    n64 = (N + 63) // 64 * 64
    out = np.zeros(n64 * K, np.int8)
    n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
    off = ((n // 64) * (K // 64) + k // 64) * 4096 + ((n // 16) % 4) * 1024 + ((k // 16) % 4) * 256 + (n % 16) * 16 + k % 16
    out[off.reshape(-1)] = W.reshape(-1)
    return out


@pytest.fixture(params=["auto", "narrow"])
def wreg_tiles(request):
    """work items of the weights-in-registers GEMM (16x16x64 form): the product's 128 x 256 tiles, or the 128 x 128 work items of
    the lab build (include/ivit_hip_debug.h flags2 bit 13: built and measured in round 4, slower, kept for A/B)"""
    if request.param == "auto":
        yield "auto"
        return
    with _lib.lab_session():
        _lib.call("ivit_debug_set_gemm_flags2", 8192)
        yield request.param


@pytest.mark.parametrize("FR", [8, 16])
@pytest.mark.parametrize("M,N,K", [(2600, 512, 384), (197 * 16, 768, 768), (2049, 192, 192), (4000, 320, 576), (197 * 12, 2304, 768),
                                   (197 * 64, 384, 384), (197 * 20, 1152, 384), (5000, 576, 192)])
def test_gemm_weight_fragment_layout(M, N, K, FR, wreg_tiles):
    """ivit_pack_weight_frags_i8 / ivit_pack_weight_frags16_i8 == the documented layouts; the weights-in-registers kernel in both
    MFMA forms (IVIT_W_FRAGS: 32x32x32, IVIT_W_FRAGS16: 16x16x64; the latter with 128 x 256 and with 128 x 128 work items) == the
    oracle for all epilogues, row-major and block-layout A, block-layout output, partial token and channel tiles"""
    if FR == 8 and wreg_tiles != "auto":
        pytest.skip("narrow tiles exist for the 16x16x64 form only")
    rng = np.random.default_rng(M + K + 1)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    md, ed = me_dev(m, e)
    dA, dW, db = dev(A), dev(W), dev(b)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dA), K, M, K, _lib.ptr(At), st())
    Wf = torch.full(((N + 63) // 64 * 64 * K,), 77, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags_i8" if FR == 8 else "ivit_pack_weight_frags16_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
    assert np.array_equal(Wf.cpu().numpy(), _frag_layout_host(W) if FR == 8 else _frag16_layout_host(W))
    exp = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    res = rng.integers(-128, 128, size=(M, N)).astype(np.int8)
    dres = dev(res)
    m1, e1 = dyadic(np.float32(0.7 * 2 ** -4), np.float32(2 ** -4))
    m2, e2 = dyadic(np.float32(2 ** -5), np.float32(2 ** -4))
    exp_res = orc.requant(exp, m1.astype(np.float64), e1, 8, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
    # a main multiplier one step below 1/2 (m = 2^30 - 1, e = 31): float32 rounds it to 1/2 and 3 * M crosses a rounding boundary,
    # so the launcher's exhaustive check rejects the float32 form of the residual QuantAct and the float64 form runs
    m3, e3 = np.array([(1 << 30) - 1], np.uint32), np.array([31], np.int32)
    exp_res64 = orc.requant(exp, m3.astype(np.float64), e3, 8, z2=res.astype(np.int32), m2=m2.astype(np.float64), e2=e2)
    for lay in (FR, FR + 1):
        a_op = At if lay & 1 else dA
        out = torch.zeros(M, N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(a_op), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(out), N, M, N, K, lay, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), exp), lay
        for (ma, ea, want) in ((m1, e1, exp_res), (m3, e3, exp_res64)):
            out = torch.zeros(M, N, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(a_op), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md),
                      _lib.ptr(ed), _lib.ptr(dres), N, int(ma[0]), int(ea[0]), int(m2[0]), int(e2[0]), _lib.ptr(out), N, M, N, K,
                      lay, st())
            assert np.array_equal(out.cpu().numpy().astype(np.int32), want), (lay, int(ma[0]))
    if M % 197 == 0 and N % 192 == 0:     # head-major q/k/v epilogue
        T, hd = 197, 64
        H = N // (3 * hd)
        outs = []
        for lay in (0, FR + 1):
            out = torch.zeros(3 * M * H * hd, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(At if lay else dA), K, _lib.ptr(Wf if lay else dW), K, _lib.ptr(db),
                      _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), T, H, hd, M, N, K, lay, st())
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
    outb = torch.zeros(R16 * N, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
              _lib.ptr(outb), N, M, N, K, FR + 5, st())
    assert np.array_equal(outb.cpu().numpy(), _block_layout_host(exp.astype(np.int8)))
    with pytest.raises(_lib.IvitError, match="fragment-packed"):   # K / 64 must be a multiple of 3
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(out), N, M, N, 128, FR, st())


@pytest.mark.parametrize("M,N,K", [(197 * 256, 768, 768), (197 * 256, 2304, 768), (197 * 64 + 77, 3072, 768), (30000, 768, 1536),
                                   (197 * 40, 1152, 3072), (4100, 256, 960)])
def test_gemm_wave_pipelined_form_equals_two_workgroup_form(M, N, K):
    """round 4 experiment, lab build only (flags2 bit 15; exact but slower): gemm_wp.h (one workgroup of eight waves
    per CU, a tile's requantisation inside the next tile's main loop, half of its accumulators parked in LDS) against the product's
    gemm_i8_wreg_kernel, which the oracle tests pin: the same bytes for the
    plain and the head-major epilogue, row-major / block-layout A, block-layout output, several tiles per workgroup (A / B accumulator
    sets alternate), partial last token tile, channel tiles beyond N, exact ties and certificate failures in the data"""
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    A[::7] = 1                                   # rows of ones: accumulators = bias + row sums of W -> many exact ties with m = 2^k
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    m[::5] = 1 << 30                             # power-of-two multipliers: exact .5 ties -> the float64 path of a unit
    md, ed = me_dev(m, e)
    dA, dW, db = dev(A), dev(W), dev(b)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dA), K, M, K, _lib.ptr(At), st())
    Wf = torch.zeros((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
    qkv = N % 192 == 0 and M % 197 == 0

    def run_all():
        outs = []
        for lay in (16, 17, 21):                 # IVIT_W_FRAGS16 | row-major A, | block A, | block A + block output
            out = torch.zeros(R16 * N, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At if lay & 1 else dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                      _lib.ptr(out), N, M, N, K, lay, st())
            outs.append(out.cpu().numpy())
        if qkv:
            hd = 64
            out = torch.zeros(M * N, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out),
                      197, N // (3 * hd), hd, M, N, K, 17, st())
            outs.append(out.cpu().numpy())
        return outs

    ref = run_all()
    with _lib.lab_session():
        _lib.call("ivit_debug_set_gemm_flags2", 32768)
        got = run_all()
    for i, (a, r) in enumerate(zip(got, ref)):
        assert np.array_equal(a, r), f"output {i}: {(a != r).sum()} of {a.size} bytes differ"
    assert np.array_equal(got[0][: M * N], got[1][: M * N])
    # and a slice against the oracle (the first 300 rows and the last 200: first and last tiles of the launch)
    rows = np.r_[0:300, M - 200:M]
    exp = orc.requant(orc.gemm_i8(A[rows], W, b), m.astype(np.float64), e, 8)
    assert np.array_equal(got[0][: M * N].reshape(M, N)[rows].astype(np.int32), exp)


@pytest.mark.parametrize("M,N,K", [(49 * 256, 288, 128), (20003, 320, 128), (9000, 96, 64), (8192, 16, 64), (49 * 200, 192, 128)])
def test_gemm_skinny_k_form(M, N, K):
    """round 4: K <= 128, N <= 320, M >= 8192 (Swin stage 0) -- the whole weight matrix in LDS, strips of 16 tokens per wave, requantisation
    on the MFMA accumulators: against the oracle (slices), against the tile kernels (lab flags2 bit 20) byte for byte, plain and
    head-major epilogue, exact ties and certificate failures, a partial last strip"""
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    A[::5] = 1
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -14, -8)
    m[::3] = 1 << 30                             # power-of-two multipliers: exact ties -> the float64 path
    md, ed = me_dev(m, e)
    dA, dW, db = dev(A), dev(W), dev(b)
    qkv = N % 96 == 0 and M % 49 == 0

    def run_all():
        out = torch.zeros(M, N + 16, dtype=torch.int8, device=DEV)         # ldo > N: the pad columns must stay untouched
        _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), N + 16, M, N, K, 0, st())
        outs = [out.cpu().numpy()]
        if qkv:
            o2 = torch.zeros(M * N, dtype=torch.int8, device=DEV)
            _lib.call("ivit_gemm_i8_requant_qkv_ex", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(o2),
                      49, N // 96, 32, M, N, K, 0, st())
            outs.append(o2.cpu().numpy())
        return outs

    got = run_all()
    with _lib.lab_session():
        _lib.call("ivit_debug_set_gemm_flags2", 1 << 20)
        ref = run_all()
        _lib.call("ivit_debug_set_gemm_flags2", 1 << 21)      # the skinny-K kernel's run-time-K instantiation (the product takes K = 64 / 128 at compile time)
        anyk = run_all()
    for i, (a, r, k) in enumerate(zip(got, ref, anyk)):
        assert np.array_equal(a, r), f"output {i}: {(a != r).sum()} of {a.size} bytes differ"
        assert np.array_equal(a, k), f"output {i}, run-time K: {(a != k).sum()} of {a.size} bytes differ"
    assert not got[0][:, N:].any()
    rows = np.r_[0:200, M - 150:M]
    exp = orc.requant(orc.gemm_i8(A[rows], W, b), m.astype(np.float64), e, 8)
    assert np.array_equal(got[0][rows, :N].astype(np.int32), exp)


def _gelu_ws(M):
    return torch.zeros((M + 127) // 128, dtype=torch.int32, device=DEV)


@pytest.fixture
def lab_library():
    """the lab build of the library (include/ivit_hip_debug.h): entry points that are not part of the product"""
    with _lib.lab_session():
        yield


@pytest.mark.parametrize("M,N,K", [(2600, 768, 192), (197 * 16, 3072, 768), (2049, 1024, 192), (4000, 1792, 576)])
def test_gemm_requant_gelu_fused(M, N, K, lab_library):
    """ivit_gemm_i8_requant_gelu_ex (lab library: the experiment of round 3, never part of an engine): the (row max, k) -> int8 table of ShiftGELU + mlp.qact1 applied inside the GEMM that produces
    k, by the workgroup that completes a 128-token panel == the oracle's GEMM + requant followed by the table with the maximum over
    the whole row; row-major and block-layout operands / output, partial token panels and channel tiles; the workspace is left zero
    and serves the next launch"""
    rng = np.random.default_rng(M + N + 11)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -17, -11)
    md, ed = me_dev(m, e)
    lut = rng.integers(-128, 128, size=(256, 256)).astype(np.int8)
    k8 = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    rmax = k8.max(axis=1)
    assert len(np.unique(rmax)) > 20 and rmax.max() <= 127          # many different table rows in play
    exp = lut[rmax[:, None] + 128, k8 + 128]
    dA, dW, db, dl = dev(A), dev(W), dev(b), dev(lut)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dA), K, M, K, _lib.ptr(At), st())
    Wf = torch.zeros((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
    ws = _gelu_ws(M)
    nb = np.zeros(1, np.int64)
    _lib.call("ivit_gemm_gelu_workspace_bytes", M, nb.ctypes.data_as(C.c_void_p))
    assert int(nb[0]) == ws.numel() * 4
    for lay in (16, 17, 16 | 4, 17 | 4, 16):
        blocks = bool(lay & 4)
        out = torch.zeros(R16 * N if blocks else M * N, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_gelu_ex", _lib.ptr(At if lay & 1 else dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md),
                  _lib.ptr(ed), _lib.ptr(dl), _lib.ptr(ws), _lib.ptr(out), N, M, N, K, lay, st())
        got = out.cpu().numpy()
        want = _block_layout_host(exp) if blocks else exp.reshape(-1)
        assert np.array_equal(got[:want.size], want), (lay, int((got[:want.size] != want).sum()))
        assert int(ws.abs().max()) == 0, lay         # left as found
    with pytest.raises(_lib.IvitError, match="IVIT_W_FRAGS16"):
        _lib.call("ivit_gemm_i8_requant_gelu_ex", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(dl), _lib.ptr(ws), _lib.ptr(out), N, M, N, K, 0, st())


def test_gemm_requant_gelu_fused_headline_shape(lab_library):
    """mlp.fc1 of DeiT-B at batch 256 (50 432 x 3072 x 768, block layouts, 394 panels of 12 channel tiles on 512 workgroups, half
    tiles in the last round) with the real ShiftGELU table: ten launches in a row on one workspace, each equal to
    ivit_gemm_i8_requant_ex + ivit_shiftgelu_lut_i8_ex in place"""
    M, N, K = 256 * 197, 3072, 768
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(DEV)
    W = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g)
    W[:, ::3] //= 8                                    # accumulators that leave a spread of row maxima after the requantisation
    W = W.to(DEV)
    b = torch.randint(-50000, 50000, (N,), dtype=torch.int32, generator=g).to(DEV)
    rng = np.random.default_rng(3)
    m, e = rand_me(rng, N, -17, -13)
    md, ed = me_dev(m, e)
    s_g = np.float32(0.0517)
    mg, eg = dyadic(np.float32(s_g * np.float32(1 / 128)), np.float32(0.011))
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_build_lut_ex", float(s_g), int(mg[0]), int(eg[0]), None, _lib.ptr(lut), st())
    At = torch.zeros(M * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), st())
    Wf = torch.zeros(N * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), st())
    ref = torch.zeros(M * N, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(ref), N,
              M, N, K, 16 | 1 | 4, st())
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(ref), N, M, N, _lib.ptr(lut), _lib.ptr(ref), N, 1 | 2, st())
    assert len(torch.unique(ref)) > 30
    ws = _gelu_ws(M)
    for it in range(10):
        out = torch.full((M * N,), 3, dtype=torch.int8, device=DEV)
        _lib.call("ivit_gemm_i8_requant_gelu_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(md), _lib.ptr(ed),
                  _lib.ptr(lut), _lib.ptr(ws), _lib.ptr(out), N, M, N, K, 16 | 1 | 4, st())
        assert torch.equal(out, ref), (it, int((out != ref).sum()))
        assert int(ws.abs().max()) == 0


@pytest.mark.parametrize("M,N,K", [(2600, 512, 384), (197 * 16, 3072, 768)])
def test_gemm_requant_output_map(M, N, K):
    """ivit_gemm_i8_requant_lut_ex: a 256-entry int8 -> int8 map applied to every requantised output in the epilogue of the
    weights-in-registers kernel (row-major and block-layout output) == the oracle's requant followed by the map"""
    rng = np.random.default_rng(M + N + 7)
    A = rng.integers(-128, 128, size=(M, K)).astype(np.int8)
    W = rng.integers(-128, 128, size=(N, K)).astype(np.int8)
    b = rng.integers(-50000, 50000, size=N).astype(np.int32)
    m, e = rand_me(rng, N, -16, -9)
    md, ed = me_dev(m, e)
    lut = rng.integers(-128, 128, size=256).astype(np.int8)
    k8 = orc.requant(orc.gemm_i8(A, W, b), m.astype(np.float64), e, 8)
    exp = lut[k8 + 128].astype(np.int32)
    dA, dW, db, dl = dev(A), dev(W), dev(b), dev(lut)
    R16 = (M + 15) // 16 * 16
    At = torch.zeros(R16 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(dA), K, M, K, _lib.ptr(At), st())
    Wf = torch.zeros((N + 63) // 64 * 64 * K, dtype=torch.int8, device=DEV)
    _lib.call("ivit_pack_weight_frags_i8", _lib.ptr(dW), K, N, K, _lib.ptr(Wf), st())
    out = torch.zeros(M, N, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_lut_ex", _lib.ptr(dA), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(dl),
              _lib.ptr(out), N, M, N, K, 8, st())
    assert np.array_equal(out.cpu().numpy().astype(np.int32), exp)
    outb = torch.zeros(R16 * N, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_lut_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(dl),
              _lib.ptr(outb), N, M, N, K, 13, st())
    assert np.array_equal(outb.cpu().numpy(), _block_layout_host(exp.astype(np.int8)))
    with pytest.raises(_lib.IvitError, match="IVIT_W_FRAGS"):
        _lib.call("ivit_gemm_i8_requant_lut_ex", _lib.ptr(dA), K, _lib.ptr(dW), K, _lib.ptr(db), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(dl),
                  _lib.ptr(out), N, M, N, K, 0, st())


def test_producers_write_block_layout():
    """LayerNorm, fused attention and the GELU table kernel with out_blocks = 1 == ivit_tile_operand_i8 of their
    row-major output (ragged row counts: the last 16-row block is partly padding)"""
    rng = np.random.default_rng(5)

    def tiled_of(rowmajor, rows, K):
        t = torch.zeros((rows + 15) // 16 * 16 * K, dtype=torch.int8, device=DEV)
        _lib.call("ivit_tile_operand_i8", _lib.ptr(rowmajor), K, rows, K, _lib.ptr(t), st())
        return t

    def valid_bytes(rows, K):   # positions of real elements in the block buffer
        X = np.ones((rows, K), np.int8)
        return _block_layout_host(X).astype(bool)

    # LayerNorm
    rows, Cn = 197 * 3 + 5, 768
    k = np.clip(np.rint(rng.normal(0, 30, size=(rows, Cn))), -128, 127).astype(np.int8)
    lp = _ln_host(rng.uniform(0.5, 1.5, size=Cn).astype(np.float32), rng.normal(0, 0.1, size=Cn).astype(np.float32), np.float32(2 ** -4))
    md, ed = me_dev(lp.m, lp.e)
    args = (_lib.ptr(dev(k)), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)), _lib.ptr(dev(lp.s_ln)), _lib.ptr(md), _lib.ptr(ed))
    rm = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i8", *args, _lib.ptr(rm), Cn, st())
    v = valid_bytes(rows, Cn)
    for form in (0, 1, 2, 3, 4):   # product library; lab: a wave per row, half a wave per row, grouped, streaming
        bl = torch.zeros((rows + 15) // 16 * 16 * Cn, dtype=torch.int8, device=DEV)
        if form == 0:
            _lib.call("ivit_layernorm_i8_ex", *args, _lib.ptr(bl), Cn, 1, st())
        else:
            with _lib.lab_session():
                _lib.call("ivit_debug_ln_wave_per_row", form)
                _lib.call("ivit_layernorm_i8_ex", *args, _lib.ptr(bl), Cn, 1, st())
        assert np.array_equal(bl.cpu().numpy()[v], tiled_of(rm, rows, Cn).cpu().numpy()[v]), form
        assert not bl.cpu().numpy()[~v].any(), form
    # GELU table form
    rows, L = 197 * 2 + 3, 3072
    x = np.clip(np.rint(rng.normal(0, 40, size=(rows, L))), -128, 127).astype(np.int8)
    mg, eg = dyadic(np.float32(0.05 / 128), np.float32(0.03))
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_build_lut", 0.05, int(mg[0]), int(eg[0]), _lib.ptr(lut), st())
    rm = torch.empty(rows, L, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(x)), L, rows, L, _lib.ptr(lut), _lib.ptr(rm), L, st())
    bl = torch.zeros((rows + 15) // 16 * 16 * L, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(dev(x)), L, rows, L, _lib.ptr(lut), _lib.ptr(bl), L, 1, st())
    v = valid_bytes(rows, L)
    assert np.array_equal(bl.cpu().numpy()[v], tiled_of(rm, rows, L).cpu().numpy()[v])
    # block-layout input, in place (how the engine chains mlp.fc1 -> GELU -> mlp.fc2); row-major in place too
    inpl = tiled_of(dev(x), rows, L)
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(inpl), L, rows, L, _lib.ptr(lut), _lib.ptr(inpl), L, 3, st())
    assert np.array_equal(inpl.cpu().numpy()[v], bl.cpu().numpy()[v])
    inpl = dev(x).clone()
    _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(inpl), L, rows, L, _lib.ptr(lut), _lib.ptr(inpl), L, 0, st())
    assert torch.equal(inpl, rm)
    with pytest.raises(_lib.IvitError, match="same layout"):
        _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(inpl), L, rows, L, _lib.ptr(lut), _lib.ptr(inpl), L, 1, st())
    # fused attention
    B, H, T, hd = 3, 6, 197, 64
    qkv = np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, hd))), -128, 127).astype(np.int8)
    ms, es = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -2))
    mo, eo = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -3))
    a = (_lib.ptr(dev(qkv)),)
    rm = torch.empty(B * T, H * hd, dtype=torch.int8, device=DEV)
    _lib.call("ivit_attention_fused_i8", a[0], _lib.ptr(rm), B, H, T, hd, int(ms[0]), int(es[0]), 0.25, int(mo[0]), int(eo[0]), st())
    bl = torch.zeros((B * T + 15) // 16 * 16 * H * hd, dtype=torch.int8, device=DEV)
    _lib.call("ivit_attention_fused_i8_ex", a[0], _lib.ptr(bl), B, H, T, hd, int(ms[0]), int(es[0]), 0.25, int(mo[0]), int(eo[0]), 1,
              st())
    v = valid_bytes(B * T, H * hd)
    assert np.array_equal(bl.cpu().numpy()[v], tiled_of(rm, B * T, H * hd).cpu().numpy()[v])


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(64, 100, dtype=torch.int8, device=DEV)
    with pytest.raises(_lib.IvitError, match="multiple of 64"):
        _lib.call("ivit_gemm_i8_i32", _lib.ptr(a), 100, _lib.ptr(a), 100, None, _lib.ptr(a), 64, 64, 64, 100, st())
    with pytest.raises(_lib.IvitError, match="NULL"):
        _lib.call("ivit_gemm_i8_i32", None, 64, _lib.ptr(a), 64, None, _lib.ptr(a), 64, 64, 64, 64, st())


# ----------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,T,p_at", [(2, 3, 197, -2), (1, 6, 197, -3), (1, 2, 208, -1), (1, 1, 193, -2),
                                        (2, 2, 192, -2), (1, 3, 145, -3), (2, 1, 101, -2), (1, 2, 50, -1), (1, 1, 17, -2), (1, 1, 5, -2)])
@pytest.mark.parametrize("s_mult", [1.0, 1.37], ids=["pow2_score_multiplier", "odd_score_multiplier"])
def test_attention_fused(B, H, T, p_at, s_mult):
    """one (image, head) per workgroup against the oracle's matmul -> requant -> Shiftmax -> matmul -> requant; token counts 193 .. 208
    take the tuned form (only the last key tile is partial), fewer tokens the general one (every key tile masked: other geometries).
    A power-of-two score multiplier takes the float32 requantisation of the scores (attention_kernel<.., RQ32>), any other the
    float64 one; scores that land exactly on .5 (ties to even) occur in both."""
    rng = np.random.default_rng(100 + B * H + T)
    hd = 64
    qkv = np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, hd))), -128, 127).astype(np.int8)
    s_a1 = np.float32(2.0 ** -4)
    s_S = np.float32(np.float32(np.float32(s_a1 * s_a1) * np.float32(0.125)) * np.float32(s_mult))
    s_at = np.float32(2.0 ** p_at)
    s_pv = np.float32(np.float32(1 / 128.0) * s_a1)
    s_a2 = np.float32(2.0 ** -3)
    ms, es = dyadic(s_S, s_at)
    mo, eo = dyadic(s_pv, s_a2)
    exp = np.empty((B, T, H * hd), np.int32)
    for b in range(B):
        for h in range(H):
            S = orc.gemm_i8(qkv[0, b, h], qkv[1, b, h])
            ka = orc.requant(S, ms.astype(np.float64), es, 8)
            P = orc.shiftmax(ka, s_at)
            assert P.max() <= 127
            O = orc.gemm_i8(P.astype(np.int8), qkv[2, b, h], transB=False)
            exp[b, :, h * hd:(h + 1) * hd] = orc.requant(O, mo.astype(np.float64), eo, 8)
    out = torch.full((B * T, H * hd), 99, dtype=torch.int8, device=DEV)
    _lib.call("ivit_attention_fused_i8", _lib.ptr(dev(qkv)), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]),
              float(s_at), int(mo[0]), int(eo[0]), st())
    got = out.cpu().numpy().astype(np.int32).reshape(B, T, H * hd)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"
    assert np.abs(exp).max() > 20


@pytest.mark.parametrize("T,band,pbits", [(197, False, 8), (197, True, 8), (203, True, 8), (193, False, 8), (197, True, 16), (193, False, 16)])
def test_attention_fused_ibert(T, band, pbits):
    """ivit_attention_fused_i8_ibert against its specification evaluated in numpy: requantised scores, table lookup over
    (row max, q), float32 row sum in torch's reduction order (oracle torch_rowsum), factor = floor(2^32 / S),
    p = floor(fl32(e * factor) / 2^(33 - pbits)), P.V (pbits = 16: ivit_attention_fused_i8_ibert_wide, p up to 2^15).  The table is synthetic -- non-integer floats, so that the summation order
    matters, and entry 16384 at distance 0 for some row maxima, so that a one-hot row gives p = 128 (the two-operand path)"""
    from ivit_amd.prepare import shiftexp_band
    rng = np.random.default_rng(T + band)
    B, H, hd = 2, 2, 64
    qkv = np.clip(np.rint(rng.normal(0, 30, size=(3, B, H, T, hd))), -128, 127).astype(np.int8)
    qkv[0, 0, 0, 5] = 0
    qkv[0, 0, 0, 5, :8] = 127                       # a query with one dominant key -> one-hot row
    qkv[1, 0, 0] = np.clip(qkv[1, 0, 0], -20, 20)
    qkv[1, 0, 0, 17, :8] = 127
    ms, es = dyadic(np.float32(2.0 ** -11), np.float32(2.0 ** -2))
    mo, eo = dyadic(np.float32(2.0 ** (-3 - pbits)), np.float32(2.0 ** -3))
    dist = np.arange(256)
    prof = np.floor(16384.0 * np.exp(-dist / 9.0))                           # exp-like, 0 beyond ~90 steps
    tab = np.zeros((256, 256), np.float32)
    for qm in range(256):
        for qq in range(qm + 1):
            v = prof[qm - qq]
            # fl(fl(k * s) / s)-like perturbation (relative 2^-22) except on the power-of-two entry that makes p = 128 reachable
            tab[qm, qq] = np.float32(v) if (qm - qq == 0 and qm % 3 == 0) else np.float32(v * (1.0 + ((qm * 7 + qq) % 5 - 2) * 2.0 ** -22))
    dtab = dev(tab.reshape(-1))
    bandt, bw = (None, 0)
    if band:
        bt, bw = shiftexp_band(tab.view(np.uint32))
        assert bw and bw <= 128
        bandt = dev(bt.view(np.float32).reshape(-1))
    exp = np.empty((B, T, H * hd), np.int32)
    n128 = 0
    for b in range(B):
        for h in range(H):
            S = orc.gemm_i8(qkv[0, b, h], qkv[1, b, h])
            ka = orc.requant(S, ms.astype(np.float64), es, 8)
            P = np.empty((T, T), np.int32)
            for i in range(T):
                e = tab[int(ka[i].max()) + 128, ka[i] + 128].astype(np.float32)
                Ssum = orc.torch_rowsum(e)
                factor = np.floor(np.float32(4294967296.0) / np.float32(Ssum)).astype(np.float32)
                P[i] = np.floor((e * factor).astype(np.float32) / np.float32(2.0 ** (33 - pbits))).astype(np.int32)
            n128 += int((P == 1 << (pbits - 1)).sum())
            assert P.max() <= 1 << (pbits - 1) and P.min() >= 0
            O = P.astype(np.int64) @ qkv[2, b, h].astype(np.int64)
            exp[b, :, h * hd:(h + 1) * hd] = orc.requant(O.astype(np.int32), mo.astype(np.float64), eo, 8)
    assert n128 > 0                                  # the p = 128 path is exercised
    out = torch.full((B * T, H * hd), 99, dtype=torch.int8, device=DEV)
    if pbits == 8:
        _lib.call("ivit_attention_fused_i8_ibert", _lib.ptr(dev(qkv)), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]), int(mo[0]),
                  int(eo[0]), _lib.ptr(dtab), _lib.ptr(bandt), bw, 0, st())
    else:
        _lib.call("ivit_attention_fused_i8_ibert_wide", _lib.ptr(dev(qkv)), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]),
                  int(mo[0]), int(eo[0]), _lib.ptr(dtab), _lib.ptr(bandt), bw, pbits, 0, st())
    got = out.cpu().numpy().astype(np.int32).reshape(B, T, H * hd)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"
    assert np.abs(exp).max() > 20


def test_attention_unsupported_geometry():
    a = torch.zeros(1 << 16, dtype=torch.int8, device=DEV)
    with pytest.raises(_lib.IvitError, match="unsupported geometry"):
        _lib.call("ivit_attention_fused_i8", _lib.ptr(a), _lib.ptr(a), 1, 1, 49, 32, 1 << 30, 40, 0.25, 1 << 30, 40, st())


# ----------------------------------------------------------------------------------- LayerNorm
def _ln_host(gamma, beta, s_out):
    from ivit_amd.prepare import LayerNormParams
    return LayerNormParams(gamma, beta, s_out)


def test_layernorm_kat(kat, ln_form):
    for ci in kat["ln_cases"][:3]:  # 8-bit inputs (the int8 kernel); case 3 is 16-bit -> i32 kernel
        c = f"ln{ci}_"
        k = kat[c + "k"]
        lp = _ln_host(kat[c + "gamma"], kat[c + "beta"], kat[c + "q_sf"])
        assert np.array_equal(lp.s_ln, kat[c + "sln"]) and np.array_equal(lp.bias_int, kat[c + "bias_int"])
        rows, Cn = k.shape
        out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
        md, ed = me_dev(lp.m, lp.e)
        _lib.call("ivit_layernorm_i8", _lib.ptr(dev(k.astype(np.int8))), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)),
                  _lib.ptr(dev(lp.s_ln)), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), Cn, st())
        got = out.cpu().numpy().astype(np.int32)
        assert np.array_equal(got, kat[c + "q_out"]), (ci, (got != kat[c + "q_out"]).sum())


def test_layernorm_f32_module_form_kat(kat):
    for ci in kat["ln_cases"]:
        c = f"ln{ci}_"
        k = kat[c + "k"]
        rows, Cn = k.shape
        out = torch.empty(rows, Cn, dtype=torch.float32, device=DEV)
        _lib.call("ivit_layernorm_i32_f32", _lib.ptr(dev(k)), Cn, rows, Cn, _lib.ptr(dev(kat[c + "bias_int"])),
                  _lib.ptr(dev(kat[c + "sln"])), _lib.ptr(out), Cn, st())
        assert np.array_equal(out.cpu().numpy().view(np.int32), kat[c + "y_bits"]), ci


@pytest.mark.parametrize("rows,Cn", [(1000, 768), (513, 192), (64, 384), (7, 1024), (20000, 768), (3000, 2048),
                                     (20001, 96), (70003, 96), (66001, 128)])   # the last three: one dword per lane (4 / 8 row pairs per wave)
def test_layernorm_random_vs_oracle(rows, Cn, ln_form):
    rng = np.random.default_rng(rows + Cn)
    k = np.clip(np.rint(rng.normal(rng.normal(0, 10, size=(rows, 1)), rng.uniform(1, 50, size=(rows, 1)),
                                   size=(rows, Cn))), -128, 127).astype(np.int8)
    gamma = rng.uniform(0.5, 1.5, size=Cn).astype(np.float32)
    beta = rng.normal(0, 0.1, size=Cn).astype(np.float32)
    y, s_ln, _ = orc.layernorm(k.astype(np.int32), gamma, beta)
    s_out = np.float32(2.0 ** np.ceil(np.log2(np.abs(y * s_ln).max() / 127 * 0.8)))
    m, e = orc.dyadic(s_ln, s_out)
    exp = orc.requant(orc.roundtrip(y, s_ln), m, e, 8)
    lp = _ln_host(gamma, beta, s_out)
    out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    md, ed = me_dev(lp.m, lp.e)
    _lib.call("ivit_layernorm_i8", _lib.ptr(dev(k)), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)),
              _lib.ptr(dev(lp.s_ln)), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), Cn, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"


@pytest.fixture(params=[0, 4, 3, 2, 1], ids=["product", "lab_streaming", "lab_grouped", "lab_half_wave_per_row", "lab_wave_per_row"])
def ln_form(request):
    """the product library's own choice of int8 LayerNorm kernel, and every form forced through the lab build: the streaming
    kernel of round 4 wherever it applies (the product takes it from ~12 MB of rows), the grouped kernel of rounds 2-3, half
    a wave per row (C <= 1536) and a wave per row"""
    if request.param == 0:
        yield 0
        return
    with _lib.lab_session():
        _lib.call("ivit_debug_ln_wave_per_row", request.param)
        yield request.param


@pytest.mark.parametrize("regime", ["tiny_gamma", "huge_gamma", "big_bias", "saturating", "vanishing", "constant_rows",
                                    "two_level_rows", "mixed_sign_gamma_free"])
@pytest.mark.parametrize("Cn", [768, 192])
def test_layernorm_certificate_regimes(regime, Cn, ln_form):
    """the float32 bracket certificate of layernorm_i8_kernel (and its literal fallback) against the oracle where the
    pieces of the chain have unusual magnitudes: |y| far below 2^22 / near 2^31, multipliers that saturate or vanish,
    degenerate rows"""
    rows = 3000
    rng = np.random.default_rng(len(regime) * 1000 + Cn)
    k = np.clip(np.rint(rng.normal(rng.normal(0, 20, size=(rows, 1)), rng.uniform(0.5, 60, size=(rows, 1)), size=(rows, Cn))),
                -128, 127).astype(np.int8)
    gamma = rng.uniform(0.5, 1.5, size=Cn).astype(np.float32)
    beta = rng.normal(0, 0.1, size=Cn).astype(np.float32)
    scale_out = 0.8
    if regime == "tiny_gamma":
        gamma = rng.uniform(1e-4, 3e-3, size=Cn).astype(np.float32)
        beta = (gamma * rng.normal(0, 0.2, size=Cn)).astype(np.float32)
    elif regime == "huge_gamma":
        gamma = rng.uniform(20, 80, size=Cn).astype(np.float32)
    elif regime == "big_bias":
        beta = rng.normal(0, 8.0, size=Cn).astype(np.float32)          # |bias_int| up to ~2^30
    elif regime == "saturating":
        scale_out = 0.02                                               # most outputs clamp at +-127 / -128
    elif regime == "vanishing":
        scale_out = 300.0                                              # outputs in {-1, 0, 1}: every product near a tie
    elif regime == "constant_rows":
        k[::3] = k[::3, :1]                                            # variance 0: the Newton chain runs on 0
    elif regime == "two_level_rows":
        k[:] = np.where(rng.random((rows, Cn)) < 0.5, -128, 127).astype(np.int8)
    y, s_ln, _ = orc.layernorm(k.astype(np.int32), gamma, beta)
    s_out = np.float32(2.0 ** np.ceil(np.log2(max(float(np.abs(y * s_ln).max()), 1e-30) / 127 * scale_out)))
    m, e = orc.dyadic(s_ln, s_out)
    exp = orc.requant(orc.roundtrip(y, s_ln), m, e, 8)
    lp = _ln_host(gamma, beta, s_out)
    out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    md, ed = me_dev(lp.m, lp.e)
    _lib.call("ivit_layernorm_i8", _lib.ptr(dev(k)), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)),
              _lib.ptr(dev(lp.s_ln)), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), Cn, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{regime}: {(got != exp).sum()} of {got.size} differ"


@pytest.mark.parametrize("Cn", [192, 768])
def test_layernorm_newton_shortcut_rows(Cn, ln_form):
    """ln_stream.h evaluates the reference's ten Newton steps (ivit_modules.py:45-49) as floor(sqrt(var)) where that is provably
    the same, and literally for rows with var < 142 883 or var + 1 a perfect square (where the recurrence has not converged / ends
    on either of two values): rows of all three kinds, the rare kind collected from a large random draw"""
    rng = np.random.default_rng(Cn)
    n = 300000 if Cn == 192 else 60000
    sig = rng.uniform(8, 45, size=(n, 1))
    k = np.clip(np.rint(rng.normal(rng.normal(0, 6, size=(n, 1)), sig, size=(n, Cn))), -128, 127).astype(np.int8)
    ki = k.astype(np.int64)
    s1 = ki.sum(1)
    mean = np.rint((s1.astype(np.float32) / np.float32(Cn))).astype(np.int64)
    var = (ki * ki).sum(1) - 2 * mean * s1 + Cn * mean * mean
    r = np.floor(np.sqrt(var.astype(np.float64))).astype(np.int64)
    special = np.nonzero((r + 1) ** 2 - 1 == var)[0]
    small = np.nonzero(var < 142883)[0][:1500]
    rest = np.nonzero((var >= 142883) & ((r + 1) ** 2 - 1 != var))[0][:1500]
    assert len(special) >= 20 and len(small) >= 100 and len(rest) >= 100, (len(special), len(small), len(rest))
    k = k[rng.permutation(np.concatenate([special, small, rest]))]
    rows = len(k)
    gamma = rng.uniform(0.5, 1.5, size=Cn).astype(np.float32)
    beta = rng.normal(0, 0.1, size=Cn).astype(np.float32)
    y, s_ln, _ = orc.layernorm(k.astype(np.int32), gamma, beta)
    s_out = np.float32(2.0 ** np.ceil(np.log2(np.abs(y * s_ln).max() / 127 * 0.8)))
    m, e = orc.dyadic(s_ln, s_out)
    exp = orc.requant(orc.roundtrip(y, s_ln), m, e, 8)
    lp = _ln_host(gamma, beta, s_out)
    out = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    md, ed = me_dev(lp.m, lp.e)
    _lib.call("ivit_layernorm_i8", _lib.ptr(dev(k)), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)),
              _lib.ptr(dev(lp.s_ln)), _lib.ptr(md), _lib.ptr(ed), _lib.ptr(out), Cn, st())
    got = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"


# ----------------------------------------------------------------------------------- GELU
def test_shiftgelu_kat_direct_and_lut(kat):
    for ci in kat["gelu_cases"]:
        c = f"gelu{ci}_"
        k = kat[c + "k"].astype(np.int8)
        s = float(kat[c + "s"])
        rows, L = k.shape
        out32 = torch.empty(rows, L, dtype=torch.int32, device=DEV)
        _lib.call("ivit_shiftgelu_i8_i32", _lib.ptr(dev(k)), L, rows, L, s, _lib.ptr(out32), L, st())
        assert np.array_equal(out32.cpu().numpy(), kat[c + "out"]), ci
        # fused requant, direct and table forms
        s_out = np.float32(2.0 ** np.ceil(np.log2(max(np.abs(kat[c + "out"]).max(), 1) * float(kat[c + "sout"]) / 127)))
        m, e = dyadic(kat[c + "sout"], s_out)
        exp = orc.requant(kat[c + "out"], m.astype(np.float64), e, 8)
        out8 = torch.empty(rows, L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_i8", _lib.ptr(dev(k)), L, rows, L, s, int(m[0]), int(e[0]), _lib.ptr(out8), L, st())
        assert np.array_equal(out8.cpu().numpy().astype(np.int32), exp), ci
        lut = torch.empty(65536, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_build_lut", s, int(m[0]), int(e[0]), _lib.ptr(lut), st())
        out8b = torch.empty(rows, L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(k)), L, rows, L, _lib.ptr(lut), _lib.ptr(out8b), L, st())
        assert np.array_equal(out8b.cpu().numpy().astype(np.int32), exp), ci


def test_shiftgelu_all_row_maxima():
    """every (row max, k) pair of the table against the oracle, incl. negative maxima (positive exp argument)"""
    s = np.float32(2.0 ** -4)
    rows = np.stack([np.minimum(np.arange(-128, 128), kmax) for kmax in range(-128, 128)]).astype(np.int8)
    exp, s_go = orc.shiftgelu(rows.astype(np.int32), s)
    m, e = dyadic(s_go, np.float32(2.0 ** -4))
    exp8 = orc.requant(exp, m.astype(np.float64), e, 8)
    out8 = torch.empty(256, 256, dtype=torch.int8, device=DEV)
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_build_lut", float(s), int(m[0]), int(e[0]), _lib.ptr(lut), st())
    _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(rows)), 256, 256, 256, _lib.ptr(lut), _lib.ptr(out8), 256, st())
    assert np.array_equal(out8.cpu().numpy().astype(np.int32), exp8)
    _lib.call("ivit_shiftgelu_i8", _lib.ptr(dev(rows)), 256, 256, 256, float(s), int(m[0]), int(e[0]),
              _lib.ptr(out8), 256, st())
    assert np.array_equal(out8.cpu().numpy().astype(np.int32), exp8)


@pytest.mark.parametrize("rows,L", [(5003, 384), (4096, 256), (4100, 128), (6000, 96), (4097, 320)])
def test_shiftgelu_table_short_rows(rows, L):
    """round 4: rows of at most 384 bytes (Swin stage 0) take half a wave each, four rows per wave; against the oracle, against the
    whole-wave-per-row form (lab bit 24), row-major and block-layout, in place"""
    rng = np.random.default_rng(rows + L)
    s = np.float32(2.0 ** -4)
    k = np.clip(np.rint(rng.normal(0, 35, size=(rows, L))), -128, 127).astype(np.int8)
    k[::7] = np.minimum(k[::7], -3)                      # rows with a negative maximum
    exp, s_go = orc.shiftgelu(k.astype(np.int32), s)
    m, e = dyadic(s_go, np.float32(2.0 ** -4))
    exp8 = orc.requant(exp, m.astype(np.float64), e, 8).astype(np.int8)
    lut = torch.empty(65536, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_build_lut", float(s), int(m[0]), int(e[0]), _lib.ptr(lut), st())
    out = torch.empty(rows, L, dtype=torch.int8, device=DEV)
    _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(k)), L, rows, L, _lib.ptr(lut), _lib.ptr(out), L, st())
    assert np.array_equal(out.cpu().numpy(), exp8)
    with _lib.lab_session():
        _lib.call("ivit_debug_ln_ablate", 1 << 24)
        out_w = torch.empty(rows, L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(k)), L, rows, L, _lib.ptr(lut), _lib.ptr(out_w), L, st())
    assert np.array_equal(out_w.cpu().numpy(), exp8)
    if L > 256:                                            # lab bit 28: the form that prefetches the next iteration's rows (many iterations per wave)
        big = np.tile(k, (14, 1))[: 4096 * 16 + 37]
        with _lib.lab_session():
            _lib.call("ivit_debug_ln_ablate", 1 << 28)
            out_p = torch.empty(big.shape[0], L, dtype=torch.int8, device=DEV)
            _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(big)), L, big.shape[0], L, _lib.ptr(lut), _lib.ptr(out_p), L, st())
        out_b = torch.empty(big.shape[0], L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(big)), L, big.shape[0], L, _lib.ptr(lut), _lib.ptr(out_b), L, st())
        exp_big = np.tile(exp8, (14, 1))[: big.shape[0]]
        assert np.array_equal(out_p.cpu().numpy(), exp_big) and np.array_equal(out_b.cpu().numpy(), exp_big)
    if L % 64 == 0:                                        # block layout in and out, in place (what the Swin engine does)
        R16 = (rows + 15) // 16 * 16
        kb = torch.zeros(R16 * L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_tile_operand_i8", _lib.ptr(dev(k)), L, rows, L, _lib.ptr(kb), st())
        _lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(kb), L, rows, L, _lib.ptr(lut), _lib.ptr(kb), L, 3, st())
        assert np.array_equal(kb.cpu().numpy()[: R16 * L], _block_layout_host(exp8)[: R16 * L])


# ----------------------------------------------------------------------------------- Shiftmax
def test_shiftmax_kat(kat):
    for ci in kat["sm_cases"]:
        c = f"sm{ci}_"
        k = kat[c + "k"].astype(np.int8)
        rows, L = k.shape
        out = torch.empty(rows, L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftmax_i8", _lib.ptr(dev(k)), L, rows, L, float(kat[c + "s"]), _lib.ptr(out), L, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), kat[c + "out"]), ci


# ----------------------------------------------------------------------------------- element-wise
def test_requant_generic_kat(kat):
    for ci in kat["rq_cases"]:
        c = f"rq{ci}_"
        z = kat[c + "z"]
        if np.abs(z).max() >= 2 ** 31:
            continue
        m, e = dyadic(kat[c + "pre"], kat[c + "zsf"])
        assert np.array_equal(m.astype(np.float64), kat[c + "m"]) and np.array_equal(e, kat[c + "e"])
        rows, Cn = z.shape
        md, ed = me_dev(m, e)
        z2p = m2d = e2d = None
        n2 = 0
        if c + "z2" in kat:
            m2, e2 = dyadic(kat[c + "pre2"], kat[c + "zsf"])
            m2d, e2d = me_dev(m2, e2)
            z2p = dev(kat[c + "z2"])
            n2 = 1
        out = torch.empty(rows, Cn, dtype=torch.int32, device=DEV)
        _lib.call("ivit_requant_i32", _lib.ptr(dev(z.astype(np.int32))), rows, Cn, _lib.ptr(md), _lib.ptr(ed), m.size,
                  _lib.ptr(z2p), _lib.ptr(m2d), _lib.ptr(e2d), n2, int(kat[c + "bits"]), _lib.ptr(out), st())
        assert np.array_equal(out.cpu().numpy(), kat[c + "out"]), ci


def test_fixedpoint_mul_and_symmetric_quant_functions_kat(kat):
    """the reference's functional names (quant_utils.py:73-119, 178-261) in quantization_utils.quant_utils: fixedpoint_mul.apply on the
    float views the reference's own call received (ops_kat.npz: one- and two-operand cases, 8 / 16 / 32 bits) and
    SymmetricQuantFunction.apply on an activation, against the reference's outputs"""
    from ivit_amd.quantization_utils import quant_utils as qu
    done = 0
    for ci in kat["rq_cases"]:
        c = f"rq{ci}_"
        z = kat[c + "z"]
        if np.abs(z).max() >= 2 ** 22:          # x = z * s must round back to z
            continue
        pre = torch.from_numpy(np.atleast_1d(kat[c + "pre"]).astype(np.float32)).to(DEV)
        x = torch.from_numpy(z.astype(np.float32)).to(DEV).view(1, *z.shape) * pre
        zsf = torch.tensor([float(kat[c + "zsf"])], device=DEV)
        ident = isf = None
        if c + "z2" in kat:
            isf = torch.tensor([float(kat[c + "pre2"])], device=DEV)
            ident = torch.from_numpy(kat[c + "z2"].astype(np.float32)).to(DEV).view(1, *z.shape) * isf
        y = qu.fixedpoint_mul.apply(x, pre, int(kat[c + "bits"]), "symmetric", zsf, ident, isf)
        assert y.dtype == torch.float32 and y.shape == x.shape
        assert np.array_equal(y.cpu().numpy().reshape(z.shape).astype(np.int32), kat[c + "out"]), ci
        done += 1
    assert done >= 4
    x, s = kat["qs_x"], kat["qs_s"]
    q = qu.SymmetricQuantFunction.apply(torch.from_numpy(x).to(DEV), 8, torch.tensor([float(s)], device=DEV), False)
    assert q.dtype == torch.float32 and np.array_equal(q.cpu().numpy().astype(np.int32).reshape(-1), kat["qs_out"].astype(np.int32).reshape(-1))
    with pytest.raises(_lib.IvitError, match="GPU"):
        qu.fixedpoint_mul.apply(torch.zeros(1, 2, 4), torch.ones(1), 8, "symmetric", torch.ones(1), None, None)


def test_quantize_and_patchify(kat):
    x = kat["qs_x"]
    s = kat["qs_s"]
    inv = float(np.float32(1.0) / s)
    out = torch.empty(x.size, dtype=torch.int8, device=DEV)
    _lib.call("ivit_quantize_input_f32_i8", _lib.ptr(dev(x)), _lib.ptr(out), x.size, inv, st())
    assert np.array_equal(out.cpu().numpy().astype(np.int32).reshape(x.shape), kat["qs_out"])
    rng = np.random.default_rng(3)
    img = rng.normal(0, 1, size=(2, 3, 224, 224)).astype(np.float32)
    A = torch.empty(2 * 196, 768, dtype=torch.int8, device=DEV)
    _lib.call("ivit_quantize_patchify_f32_i8", _lib.ptr(dev(img)), _lib.ptr(A), 2, 3, 224, 16, inv, st())
    k0 = orc.quant_sym(img, s, 8)
    exp = k0.reshape(2, 3, 14, 16, 14, 16).transpose(0, 2, 4, 1, 3, 5).reshape(2 * 196, 768)
    assert np.array_equal(A.cpu().numpy().astype(np.int32), exp)


def test_head_argmax():
    rng = np.random.default_rng(4)
    acc = rng.integers(-100000, 100000, size=(9, 1000)).astype(np.int32)
    acc[3, 17] = acc[3, 900] = 10 ** 6  # tie on the raw accumulator, broken by the per-class scale
    s = rng.uniform(1e-6, 2e-6, size=1000).astype(np.float32)
    lf = torch.empty(9, 1000, dtype=torch.float32, device=DEV)
    t1 = torch.empty(9, dtype=torch.int32, device=DEV)
    _lib.call("ivit_head_argmax", _lib.ptr(dev(acc)), _lib.ptr(dev(s)), 9, 1000, _lib.ptr(lf), _lib.ptr(t1), st())
    exp = (acc.astype(np.float32) * s[None]).astype(np.float32)
    assert np.array_equal(lf.cpu().numpy().view(np.int32), exp.view(np.int32))
    assert np.array_equal(t1.cpu().numpy(), np.argmax(exp, axis=1))
