"""GPU parity in the NATURAL-scale regime (activation ranges as calibrated: the regime of a real checkpoint).  The
reference's I-LayerNorm / ShiftGELU / Shiftmax then see phi_s(q) = fl(fl(q*s)/s) instead of q; the engine carries that
as 256-entry tables in front of the same integer kernels (prepare.py, include/ivit_hip.h `*_compat`).  Checked against
known-answer vectors produced by the reference's own modules (compat_kat.npz), against the compat oracle, and end to
end against the reference's logits on un-snapped DeiT-T / DeiT-S (deit_*_natural.npz)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ivit = pytest.importorskip("ivit_amd")
from ivit_amd import _lib, synth  # noqa: E402
from ivit_amd.checkpoint import load_synthetic_model  # noqa: E402
from ivit_amd.engine import IntViTEngine  # noqa: E402
from ivit_amd.prepare import LayerNormParams, dyadic, phi_tables, shiftexp2d, sym_scale  # noqa: E402

DEV = "cuda:0"
_KEEP = []


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def st():
    return _lib.stream_ptr()


@pytest.fixture(scope="module")
def ckat(golden_dir):
    return np.load(os.path.join(golden_dir, "compat_kat.npz"))


@pytest.fixture(params=[0, 4, 3], ids=["product", "lab_streaming", "lab_grouped"])
def ln_form(request):
    """the product library's own choice of LayerNorm kernel, and the streaming / grouped kernels forced through the lab build"""
    if request.param == 0:
        yield 0
        return
    with _lib.lab_session():
        _lib.call("ivit_debug_ln_wave_per_row", request.param)
        yield request.param


def _ln_compat(q, s_in, gamma, beta, s_out, blocks=0):
    lp = LayerNormParams(gamma, beta, s_out)
    remap, phi = phi_tables(s_in)
    rows, Cn = q.shape
    out = torch.empty((rows + 15) // 16 * 16, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i8_compat", _lib.ptr(dev(q.astype(np.int8))), Cn, rows, Cn, _lib.ptr(dev(lp.bias_int)),
              _lib.ptr(dev(lp.s_ln)), _lib.ptr(dev(lp.m.view(np.int32))), _lib.ptr(dev(lp.e)), _lib.ptr(dev(remap)),
              _lib.ptr(dev(phi)), _lib.ptr(out), Cn, blocks, st())
    return out[:rows].cpu().numpy().astype(np.int32)


def test_layernorm_compat_kat(ckat, ln_form):
    """the reference's IVITIntLayerNorm + QuantAct on q*s inputs; every other row is an exact .5 tie of the mean"""
    for ci in ckat["ln_cases"]:
        c = f"ln{ci}_"
        lo, hi = ckat[c + "range"]
        got = _ln_compat(ckat[c + "q"], ckat[c + "s"], ckat[c + "gamma"], ckat[c + "beta"], sym_scale(lo, hi, 8))
        assert np.array_equal(got, ckat[c + "q_out"]), (ci, int((got != ckat[c + "q_out"]).sum()))


@pytest.mark.parametrize("rows,Cn,s", [(4000, 768, 0.0371), (3001, 192, 0.11873), (999, 384, 0.0052341), (50, 1024, 0.3127),
                                        (2000, 1536, 0.0371), (300, 96, 0.0371)])
def test_layernorm_compat_random_vs_oracle(rows, Cn, s, ln_form):
    rng = np.random.default_rng(rows + Cn)
    s = np.float32(s)
    q = np.clip(np.rint(rng.normal(rng.normal(0, 10, size=(rows, 1)), rng.uniform(1, 50, size=(rows, 1)),
                                   size=(rows, Cn))), -128, 127).astype(np.int32)
    for r in range(0, rows, 3):        # a third of the rows: exact ties of the mean (decided by the reduction order)
        tgt = Cn // 2 + Cn * int(rng.integers(-10, 10))
        d = tgt - int(q[r].sum())
        idx = rng.permutation(Cn)
        for c in idx:
            if d == 0:
                break
            nv = int(np.clip(q[r, c] + d, -128, 127))
            d -= nv - q[r, c]
            q[r, c] = nv
    gamma = rng.uniform(0.5, 1.5, size=Cn).astype(np.float32)
    beta = rng.normal(0, 0.1, size=Cn).astype(np.float32)
    y, s_ln, _, ties = orc.layernorm_compat(q, s, gamma, beta)
    assert ties >= rows // 4
    s_out = np.float32(np.abs(y * s_ln).max() / 127 * 0.83)
    m, e = orc.dyadic(s_ln, s_out)
    exp = orc.requant(orc.roundtrip(y, s_ln), m, e, 8)
    got = _ln_compat(q, s, gamma, beta, s_out)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"
    # the plain kernel (no phi) is a different function at this scale: the test has teeth
    y0, s0, _ = orc.layernorm(q, gamma, beta)
    assert not np.array_equal(orc.requant(orc.roundtrip(y0, s0), m, e, 8), exp)


@pytest.mark.parametrize("rows,Cn,outer,s", [(6272, 96, 3136, 0.0371), (4096, 64, 64, 0.0213), (1280, 192, 320, 0.0371), (3136, 96, 49, 0.0371)])
def test_layernorm_compat_outer_order_8_bit_equals_16_bit_kernel(rows, Cn, outer, s):
    """Swin's patch-embed LayerNorm at natural scales (8-bit input, the mean over a TRANSPOSED view: torch's outer-reduction order,
    IVIT_LN_OUTER_MEAN).  Round 4 sums a candidate tie row's groups of 16 on one lane each instead of serially; the 16-bit kernel of
    csrc/swin.hip computes the same LayerNorm from the same integers with its own outer-order sums (pinned by the reference's Swin
    goldens and tests/test_gpu_swin.py): identical bytes, on rows of which half are exact ties (outer = 49: the serial form, tail
    columns)"""
    from ivit_amd.prepare import markstein_division_ok
    rng = np.random.default_rng(rows + Cn)
    s = np.float32(s)
    tabs = phi_tables(s)
    assert tabs is not None and markstein_division_ok(s, 16)
    remap, phi = tabs
    q = np.clip(np.rint(rng.normal(rng.normal(0, 20, size=(rows, 1)), rng.uniform(5, 40, size=(rows, 1)), size=(rows, Cn))), -128, 127).astype(np.int32)
    for r in range(0, rows, 2):
        d = Cn // 2 + Cn * int(rng.integers(-20, 20)) - int(q[r].sum())
        for c in rng.permutation(Cn):
            if d == 0:
                break
            nv = int(np.clip(q[r, c] + d, -128, 127))
            d -= nv - q[r, c]
            q[r, c] = nv
    lp = LayerNormParams(rng.uniform(0.5, 1.5, size=Cn).astype(np.float32), rng.normal(0, 0.1, size=Cn).astype(np.float32), np.float32(0.031))
    b, sl, m, e = dev(lp.bias_int), dev(lp.s_ln), dev(lp.m.view(np.int32)), dev(lp.e)
    out8 = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i8_compat", _lib.ptr(dev(q.astype(np.int8))), Cn, rows, Cn, _lib.ptr(b), _lib.ptr(sl), _lib.ptr(m), _lib.ptr(e),
              _lib.ptr(dev(remap)), _lib.ptr(dev(phi)), _lib.ptr(out8), Cn, outer << 8, st())
    out16 = torch.empty(rows, Cn, dtype=torch.int8, device=DEV)
    _lib.call("ivit_layernorm_i16_i8_compat", _lib.ptr(dev(q.astype(np.int16))), rows, Cn, float(s), 1 | (outer << 8), _lib.ptr(b), _lib.ptr(sl),
              _lib.ptr(m), _lib.ptr(e), _lib.ptr(out16), Cn, 0, 0, 0, 0, st())
    a, c = out8.cpu().numpy(), out16.cpu().numpy()
    assert np.array_equal(a, c), f"{(a != c).sum()} of {a.size} bytes differ"


def test_shiftgelu_compat_table(ckat):
    for ci in ckat["gelu_cases"]:
        c = f"gelu{ci}_"
        q, s = ckat[c + "q"], np.float32(ckat[c + "s"])
        ref = ckat[c + "out"]                                   # the reference module's k' * sigmoid_int
        s_go = np.float32(s * np.float32(1 / 128.0))
        s_out = np.float32(np.abs(ref).max() * float(s_go) / 127 * 0.9)
        m, e = dyadic(s_go, s_out)
        exp = orc.requant(ref, m.astype(np.float64), e, 8)
        remap, _ = phi_tables(s)
        lut = torch.empty(65536, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_build_lut_ex", float(s), int(m[0]), int(e[0]), _lib.ptr(dev(remap)), _lib.ptr(lut), st())
        rows, L = q.shape
        out = torch.empty(rows, L, dtype=torch.int8, device=DEV)
        _lib.call("ivit_shiftgelu_lut_i8", _lib.ptr(dev(q)), L, rows, L, _lib.ptr(lut), _lib.ptr(out), L, st())
        assert np.array_equal(out.cpu().numpy().astype(np.int32), exp), ci


@pytest.mark.parametrize("B,H,T,s_at", [(2, 3, 197, 0.3127), (1, 6, 197, 0.11873), (1, 2, 208, 0.9113), (1, 1, 193, 0.2113),
                                        (1, 2, 197, 0.0571), (1, 2, 101, 0.2113), (2, 1, 50, 0.11873)])
@pytest.mark.parametrize("form", ["band_in_lds", "full_table_gather"])
def test_attention_fused_compat(B, H, T, s_at, form):
    rng = np.random.default_rng(200 + B * H + T)
    hd = 64
    qkv = np.clip(np.rint(rng.normal(0, 40, size=(3, B, H, T, hd))), -128, 127).astype(np.int8)
    s_a1 = np.float32(0.0571)
    s_S = np.float32(np.float32(s_a1 * s_a1) * np.float32(0.125))
    s_at = np.float32(s_at)
    s_pv = np.float32(np.float32(1 / 128.0) * s_a1)
    s_a2 = np.float32(0.1173)
    ms, es = dyadic(s_S, s_at)
    mo, eo = dyadic(s_pv, s_a2)
    exp = np.empty((B, T, H * hd), np.int32)
    for b in range(B):
        for h in range(H):
            S = orc.gemm_i8(qkv[0, b, h], qkv[1, b, h])
            ka = orc.requant(S, ms.astype(np.float64), es, 8)
            P = orc.shiftmax_compat(ka, s_at)
            O = orc.gemm_i8(P.astype(np.int8), qkv[2, b, h], transB=False)
            exp[b, :, h * hd:(h + 1) * hd] = orc.requant(O, mo.astype(np.float64), eo, 8)
    out = torch.full((B * T, H * hd), 99, dtype=torch.int8, device=DEV)
    tab2d = shiftexp2d(s_at)
    if form == "band_in_lds":
        from ivit_amd.prepare import shiftexp_band
        band, bw = shiftexp_band(tab2d)
        assert 16 <= bw <= 256
        _lib.call("ivit_attention_fused_i8_compat_band", _lib.ptr(dev(qkv)), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]),
                  float(s_at), int(mo[0]), int(eo[0]), None, _lib.ptr(dev(band.view(np.int32))), bw, 0, st())
    else:
        _lib.call("ivit_attention_fused_i8_compat", _lib.ptr(dev(qkv)), _lib.ptr(out), B, H, T, hd, int(ms[0]), int(es[0]),
                  float(s_at), int(mo[0]), int(eo[0]), _lib.ptr(dev(tab2d.view(np.int32))), 0, st())
    got = out.cpu().numpy().astype(np.int32).reshape(B, T, H * hd)
    assert np.array_equal(got, exp), f"{(got != exp).sum()} of {got.size} differ"
    assert np.abs(exp).max() > 5


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.int32).tobytes())


@pytest.mark.parametrize("tag,batch", [("deit_tiny_natural", 8), ("deit_small_natural", 4), ("deit_base_natural", 2), ("vit_large_natural", 2)])
def test_natural_scale_model_matches_reference(tag, batch):
    """un-snapped calibration ranges: INT32 logits, top-1 and every materialised tap equal the reference's"""
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=batch)
    assert eng.natural_sites >= 3 * cfg["depth"]          # the compat tables really are in play
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    taps = {}
    li, lf, t1 = eng.forward(imgs, taps)
    torch.cuda.synchronize()
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    bad = [n for n, t in taps.items() if n in gold and _crc(t.cpu().numpy().astype(np.int32)) != int(gold[n])]
    assert not bad, bad[:6]
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy().astype(np.int64), z["top1"])
    # the reference's float logits are an sgemm over phi values: equal to acc * scale to a small fraction of one integer step
    ref_f = z["logits_f32_bits"].view(np.float32)
    assert np.abs(lf.cpu().numpy() - ref_f).max() < 0.05 * np.abs(ref_f).max() / max(1, np.abs(z["logits_int32"]).max())


def test_natural_scale_large_batch_block_layout():
    """batch 16 > the persistent-GEMM threshold: block-layout producers (LN compat writes blocks) at natural scales"""
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny_natural")
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=32)
    gold = synth.make_images(meta["n_images"], meta["image_seed"])
    imgs = synth.make_images(32, 4242)
    pos = [0, 5, 9, 13, 17, 21, 30, 31]
    for p, g in zip(pos, gold):
        imgs[p] = g
    li, _, t1 = eng.forward(torch.from_numpy(imgs).to(DEV))
    assert np.array_equal(li.cpu().numpy()[pos], z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy()[pos].astype(np.int64), z["top1"])


# ----------------------------------------------------------------------------------- module path (literal float kernels)
def test_literal_module_kernels_kat(ckat):
    """the nn.Module mirrors at natural scales: float view q*s in, the reference module's float output bit for bit"""
    import ivit_amd.quantization_utils as qu
    for ci in ckat["ln_cases"]:
        c = f"ln{ci}_"
        qv, s = ckat[c + "q"].astype(np.float32), np.float32(ckat[c + "s"])
        Cn = qv.shape[1]
        ln = qu.IVITIntLayerNorm(Cn).to(DEV)
        ln.weight.data = torch.from_numpy(ckat[c + "gamma"]).to(DEV)
        ln.bias.data = torch.from_numpy(ckat[c + "beta"]).to(DEV)
        x = torch.from_numpy((qv * s).astype(np.float32)[None]).to(DEV)
        y, _ = ln(x, torch.tensor([float(s)], device=DEV))
        assert np.array_equal(y.cpu().numpy().reshape(-1, Cn).view(np.int32), ckat[c + "y_bits"]), ci
    sm = qu.IVITIntSoftmax().to(DEV)
    for ci in ckat["sm_cases"]:
        c = f"sm{ci}_"
        qv, s = ckat[c + "q"].astype(np.float32), np.float32(ckat[c + "s"])
        x = torch.from_numpy((qv * s).astype(np.float32)[None, None]).to(DEV)
        y, so = sm(x, torch.tensor([float(s)], device=DEV))
        got = torch.round(y / so).cpu().numpy().reshape(qv.shape).astype(np.int32)
        assert np.array_equal(got, ckat[c + "out"]), ci
    g = qu.IVITIntGELU().to(DEV)
    for ci in ckat["gelu_cases"]:
        c = f"gelu{ci}_"
        qv, s = ckat[c + "q"].astype(np.float32), np.float32(ckat[c + "s"])
        x = torch.from_numpy((qv * s).astype(np.float32)[None]).to(DEV)
        y, so = g(x, torch.tensor([float(s)], device=DEV))
        got = torch.round(y / so).cpu().numpy().reshape(qv.shape).astype(np.int32)
        assert np.array_equal(got, ckat[c + "out"]), ci


def test_natural_scale_module_path_equals_engine_and_reference():
    """the frozen nn.Module with ranges as calibrated: module-by-module path (literal float kernels) and fused engine
    (phi tables) give the same INT32 logits as the reference"""
    import ivit_amd.quantization_utils as qu
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny_natural")
    model = ivit.deit_tiny_patch16_224()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, qu.QuantAct):
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(3, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        ye = model(imgs)
        assert model._engine is not None and model._engine[2].natural_sites > 0
        model.use_engine = False
        ym = model(imgs)
    s_head = model._engine[2].head_scale[:1000].cpu().numpy()
    for y in (ye, ym):
        li = np.rint(y.cpu().numpy().astype(np.float64) / s_head).astype(np.int32)
        assert np.array_equal(li, z["logits_int32"][:3])
    assert np.array_equal(ye.cpu().numpy().view(np.int32), ym.cpu().numpy().view(np.int32))


def test_natural_scale_headline_batch_256_deit_base():
    """config 3 at full size in the regime of a real checkpoint: the reference's un-snapped DeiT-B images embedded in a batch
    of 256 (persistent GEMMs, block-layout operands, LayerNorm v2 COMPAT writing blocks, attention with the 2-D table)
    reproduce its INT32 logits; HIP-graph replay equals the eager forward"""
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_base_natural")
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=256)
    assert eng.natural_sites >= 3 * cfg["depth"]
    imgs = synth.make_images(256, 909)
    gold = synth.make_images(meta["n_images"], meta["image_seed"])
    pos = [1, 254]
    for p, g in zip(pos, gold):
        imgs[p] = g
    x = torch.from_numpy(imgs).to(DEV)
    li, _, t1 = eng.forward(x)
    li = li.cpu().numpy().copy()
    assert np.array_equal(li[pos], z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy()[pos].astype(np.int64), z["top1"])
    lg, _, _ = eng.forward_graph(x, resident=True)
    assert np.array_equal(lg.cpu().numpy(), li)


def test_vit_large_width_calibrated_on_gpu_three_way():
    """ViT-L channel width (C = 1024, 16 heads; 3 blocks to keep it short): the ranges come from two calibration forwards of
    the module mirror on the GPU (running min/max + EMA, i.e. NATURAL scales, no snapping); then the fused engine (phi
    tables, LayerNorm v2 with four dwords per lane), the module-by-module path (literal float kernels) and the CPU compat
    oracle -- three independent implementations -- agree on the INT32 logits."""
    import ivit_amd.quantization_utils as qu
    from ivit_amd.vit_quant import VisionTransformer
    depth = 3
    fs = synth.make_float_state("vit_large_patch16_224", 44, depth=depth)
    model = VisionTransformer(embed_dim=1024, depth=depth, num_heads=16)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    assert not unexpected
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(2, 771)).to(DEV))
        model(torch.from_numpy(synth.make_images(2, 772)).to(DEV))
    ivit.freeze_model(model)
    ranges = {n: (np.float32(float(m.x_min)), np.float32(float(m.x_max))) for n, m in model.named_modules()
              if isinstance(m, qu.QuantAct)}
    imgs_np = synth.make_images(2, 773)
    imgs = torch.from_numpy(imgs_np).to(DEV)
    with torch.no_grad():
        ye = model(imgs)
        eng = model._engine[2]
        assert eng.natural_sites > 0 and eng.C == 1024
        li_e = eng.forward(imgs)[0].cpu().numpy().copy()
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(ye.cpu().numpy().view(np.int32), ym.cpu().numpy().view(np.int32))
    om = orc.OracleViT(fs, ranges, 1024, depth, 16, compat=True)
    assert np.array_equal(li_e, om.forward(imgs_np)["logits_int32"])
