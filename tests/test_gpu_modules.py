"""The nn.Module mirror of the reference's operator API (i-vit_amd/quantization_utils, vit_quant.py) on the GPU:
module outputs are the reference's float32 `integer * scale` tensors, compared BITWISE with vectors the reference's
own modules produced (tests/golden/ops_kat.npz) and with the whole-model goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ivit = pytest.importorskip("ivit_amd")
from ivit_amd import _lib, synth  # noqa: E402
from ivit_amd.checkpoint import load_synthetic_model  # noqa: E402
from ivit_amd.prepare import LayerNormParams  # noqa: E402
import ivit_amd.quantization_utils as q  # noqa: E402

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits(x):
    return x.detach().cpu().numpy().view(np.int32)


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_kat.npz"))


def frozen_qact(bits_, s_out):
    qa = q.QuantAct(bits_).to(DEV)
    n = 2 ** (bits_ - 1) - 1
    qa.x_max.fill_(float(s_out) * n)
    qa.x_min.fill_(-float(s_out) * n)
    qa.fix()
    return qa


def test_quantact_kat(kat):
    for ci in kat["rq_cases"]:
        c = f"rq{ci}_"
        z, pre = kat[c + "z"], kat[c + "pre"]
        if np.abs(z).max() >= 2 ** 31 or kat[c + "zsf"] < np.finfo(np.float32).eps:
            continue  # (QuantAct clamps its scale to float32 eps; that vector drives fixedpoint_mul directly)
        qa = frozen_qact(int(kat[c + "bits"]), kat[c + "zsf"])
        x = (t(z.astype(np.float32)) * t(pre)).unsqueeze(0)           # the producer's float view
        kw = {}
        if c + "z2" in kat:
            kw = dict(identity=(t(kat[c + "z2"].astype(np.float32)) * float(kat[c + "pre2"])).unsqueeze(0),
                      identity_scaling_factor=t(np.array([kat[c + "pre2"]], np.float32)))
        y, s = qa(x, t(pre), **kw)
        assert float(s) == float(kat[c + "zsf"])
        exp = (kat[c + "out"].astype(np.float32) * kat[c + "zsf"]).astype(np.float32)
        assert np.array_equal(bits(y)[0], exp.view(np.int32)), ci


def test_quantact_input_mode(kat):
    qa = q.QuantAct().to(DEV)
    qa.x_max.fill_(4.0)
    qa.x_min.fill_(-4.0)
    qa.fix()
    y, s = qa(t(kat["qs_x"]))
    assert float(s) == float(kat["qs_s"])
    assert np.array_equal(bits(y), (kat["qs_out"].astype(np.float32) * kat["qs_s"]).astype(np.float32).view(np.int32))


def test_quantlinear_kat(kat):
    lin = q.QuantLinear(80, 24)
    pad = q.QuantLinear(128, 24)     # the HIP GEMM wants K % 64 == 0: zero-pad K (no effect on the products)
    W = np.zeros((24, 128), np.float32)
    W[:, :80] = kat["lin_W"]
    pad.weight.data = torch.from_numpy(W)
    pad.bias.data = torch.from_numpy(kat["lin_b"])
    pad.to(DEV)
    x = np.zeros((1, 10, 128), np.float32)
    x[0, :, :80] = kat["lin_x"]
    s_in = kat["lin_sin"]
    y, s = pad(t(x) * float(s_in), t(np.array([s_in], np.float32)))
    assert np.array_equal(s.cpu().numpy(), kat["lin_sacc"])
    assert np.array_equal(pad.weight_integer.cpu().numpy()[:, :80].astype(np.int32), kat["lin_wint"])
    assert np.array_equal(pad.bias_integer.cpu().numpy().astype(np.int32), kat["lin_bint"])
    exp = (kat["lin_acc"].astype(np.float32) * kat["lin_sacc"][None]).astype(np.float32)
    assert np.array_equal(bits(y)[0], exp.view(np.int32))
    assert lin.state_dict().keys() == pad.state_dict().keys()


def test_quantconv_kat(kat):
    conv = q.QuantConv2d(3, 8, kernel_size=16, stride=16)
    conv.weight.data = torch.from_numpy(kat["conv_W"])
    conv.bias.data = torch.from_numpy(kat["conv_b"])
    conv.to(DEV)
    s_in = kat["conv_sin"]
    y, s = conv(t(kat["conv_x"].astype(np.float32)) * float(s_in), t(np.array([s_in], np.float32)))
    assert s.shape == (1, 8, 1, 1) and np.array_equal(s.reshape(-1).cpu().numpy(), kat["conv_sacc"])
    exp = (kat["conv_acc"].astype(np.float32) * kat["conv_sacc"].reshape(1, 8, 1, 1)).astype(np.float32)
    assert np.array_equal(bits(y), exp.view(np.int32))


def test_quantmatmul_kat(kat):
    mm = q.QuantMatMul().to(DEV)
    sa = 2.0 ** -4
    s1 = t(np.array([sa], np.float32))
    y, s = mm(t(kat["mm_a"].astype(np.float32)) * sa, s1, t(kat["mm_b"].astype(np.float32)) * sa, s1)
    assert np.array_equal(s.cpu().numpy(), kat["mm_s"])
    exp = (kat["mm_out"].astype(np.float32) * kat["mm_s"]).astype(np.float32)
    assert np.array_equal(bits(y), exp.view(np.int32))
    # the q.k^T call pattern of vit_quant.py:72-73 (transposed, non-contiguous second operand)
    kT = (t(kat["mm_b"].astype(np.float32)) * sa).transpose(-2, -1).contiguous().transpose(-2, -1)
    y2, _ = mm(t(kat["mm_a"].astype(np.float32)) * sa, s1, kT, s1)
    assert np.array_equal(bits(y2), exp.view(np.int32))


def test_layernorm_module_kat(kat):
    for ci in kat["ln_cases"]:
        c = f"ln{ci}_"
        k = kat[c + "k"]
        ln = q.IVITIntLayerNorm(k.shape[1])
        ln.weight.data = torch.from_numpy(kat[c + "gamma"])
        ln.bias.data = torch.from_numpy(kat[c + "beta"])
        ln.to(DEV)
        s = float(kat[c + "s"])
        y, sln = ln((t(k.astype(np.float32)) * s).unsqueeze(0), t(np.array([s], np.float32)))
        assert np.array_equal(sln.cpu().numpy(), kat[c + "sln"])
        assert np.array_equal(bits(y)[0], kat[c + "y_bits"]), ci
        assert np.array_equal(ln.bias_integer.cpu().numpy(), kat[c + "bias_int"])
        if ci < 3:  # followed by the 8-bit QuantAct, as in Block.forward
            qa = frozen_qact(8, kat[c + "q_sf"])
            yq, sq = qa(y, sln)
            exp = (kat[c + "q_out"].astype(np.float32) * kat[c + "q_sf"]).astype(np.float32)
            assert np.array_equal(bits(yq)[0], exp.view(np.int32)), ci


def test_gelu_softmax_modules_kat(kat):
    for ci in kat["gelu_cases"]:
        c = f"gelu{ci}_"
        g = q.IVITIntGELU().to(DEV)
        s = float(kat[c + "s"])
        y, so = g((t(kat[c + "k"].astype(np.float32)) * s).unsqueeze(0), t(np.array([s], np.float32)))
        assert float(so) == float(kat[c + "sout"])
        exp = (kat[c + "out"].astype(np.float32) * kat[c + "sout"]).astype(np.float32)
        assert np.array_equal(bits(y)[0], exp.view(np.int32)), ci
    for ci in kat["sm_cases"]:
        c = f"sm{ci}_"
        sm = q.IVITIntSoftmax().to(DEV)
        s = float(kat[c + "s"])
        y, so = sm((t(kat[c + "k"].astype(np.float32)) * s).reshape(1, 1, *kat[c + "k"].shape),
                   t(np.array([s], np.float32)))
        assert float(so) == 1 / 128
        exp = (kat[c + "out"].astype(np.float32) / 128).astype(np.float32)
        assert np.array_equal(bits(y)[0, 0], exp.view(np.int32)), ci


def test_registry_and_aliases():
    assert q.get_gelu("ivit") is q.IVITIntGELU and q.get_softmax("ivit") is q.IVITIntSoftmax
    assert q.get_layernorm("ivit") is q.IVITIntLayerNorm and q.IntGELU is q.IVITIntGELU
    assert q.get_gelu("ibert") is q.IBERTIntGELU and q.get_layernorm("ibert") is q.IBERTIntLayerNorm
    with pytest.raises(KeyError, match="only"):
        q.get_gelu("ppoly")


def load_model(tag):
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    fam = meta.get("family", "ivit")
    kw = dict(gelu_type=fam, softmax_type=fam, layernorm_type=fam) if fam != "ivit" else {}
    model = getattr(ivit, meta["factory"])(**kw)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    assert not unexpected
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct) and name in ranges:   # Swin's act_out is never called and has no range
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    return model, meta, z


def test_model_module_path_matches_reference_golden():
    """DeiT-T through the module-by-module path (reference call protocol), 2 golden images: float logits bitwise."""
    model, meta, z = load_model("deit_tiny")
    model.use_engine = False
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"][:2])
    # buffers the reference leaves in the state_dict after a forward (Appendix D) are populated
    sd = model.state_dict()
    assert sd["blocks.0.attn.qkv.weight_integer"].abs().max() > 0 and sd["blocks.3.qact1.act_scaling_factor"].item() > 0


def test_model_engine_path_matches_reference_golden():
    """the same frozen nn.Module, default path = fused int8 engine"""
    model, meta, z = load_model("deit_small")
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"])


def test_calibration_then_freeze_runs():
    """running-stat mode (calibration forward) followed by freeze: the protocol of scripts/inference.py:33-91,223"""
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny")
    model = ivit.deit_tiny_patch16_224().to(DEV)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.eval()
    imgs = torch.from_numpy(synth.make_images(2, 7)).to(DEV)
    with torch.no_grad():
        model(imgs)                       # calibration: initialises every x_min / x_max
    assert all(float(m.x_max) > 0 for m in model.modules() if isinstance(m, q.QuantAct))
    ivit.freeze_model(model)
    assert model.is_frozen()


# ----------------------------------------------------------------------------------- Swin module mirror
@pytest.mark.parametrize("tag", ["swin_tiny", "swin_small"])
def test_swin_module_path_matches_reference_golden(tag):
    """Swin-T / Swin-S through the module-by-module path (reference call protocol: float views, torch roll / partition /
    mask-add between the HIP-backed modules), 1 golden image: float logits bitwise."""
    model, meta, z = load_model(tag)
    model.use_engine = False
    imgs = torch.from_numpy(synth.make_images(1, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"][:1])
    sd = model.state_dict()
    assert sd["layers.0.downsample.reduction.weight_integer"].abs().max() > 0
    assert "layers.0.downsample.reduction.bias_integer" not in sd
    assert sd["layers.1.blocks.1.attn.qact4.act_scaling_factor"].item() > 0


@pytest.mark.parametrize("tag", ["swin_tiny", "swin_small"])
def test_swin_engine_path_matches_reference_golden(tag):
    """the same frozen nn.Module, default path = fused integer engine"""
    model, meta, z = load_model(tag)
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"])


def test_swin_calibration_then_freeze_runs():
    fs, ranges, cfg, meta, z = load_synthetic_model("swin_tiny")
    model = ivit.swin_tiny_patch4_window7_224().to(DEV)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.eval()
    imgs = torch.from_numpy(synth.make_images(1, 7)).to(DEV)
    with torch.no_grad():
        model(imgs)
    called = [n for n, m in model.named_modules() if isinstance(m, q.QuantAct) and float(m.x_max) > 0]
    assert called == synth.swin_qact_names(cfg["depths"])      # act_out is constructed but never called
    ivit.freeze_model(model)
    assert model.is_frozen()


# ----------------------------------------------------------------------------------- calibration + harness (rows f1, f4)
def test_minmax_kernel():
    rng = np.random.default_rng(3)
    for n in (1, 63, 1000, 3 * 224 * 224 * 4 + 1):
        x = rng.standard_normal(n).astype(np.float32) * 7
        x[rng.integers(0, n)] = -123.5
        t = torch.from_numpy(x).to(DEV)
        mm = torch.empty(2, dtype=torch.float32, device=DEV)
        _lib.call("ivit_minmax_f32", _lib.ptr(t), n, _lib.ptr(mm), _lib.stream_ptr())
        assert mm.cpu().numpy().tolist() == [float(x.min()), float(x.max())]
    t = torch.tensor([float("nan"), -0.0, 0.0, float("nan")], device=DEV)
    mm = torch.empty(2, dtype=torch.float32, device=DEV)
    _lib.call("ivit_minmax_f32", _lib.ptr(t), 4, _lib.ptr(mm), _lib.stream_ptr())
    assert mm.cpu().numpy().tolist() == [0.0, 0.0] and np.signbit(mm.cpu().numpy()[0])


def test_quantact_running_stat_matches_reference_update_rule():
    """first call initialises the range, later calls apply the EMA (quant_modules.py:346-360)"""
    qa = q.QuantAct().to(DEV)
    a = torch.linspace(-3.0, 5.0, 1001, device=DEV).reshape(1, -1)
    b = torch.linspace(-7.0, 1.0, 1001, device=DEV).reshape(1, -1)
    qa(a, torch.ones(1, device=DEV))
    assert (float(qa.x_min), float(qa.x_max)) == (-3.0, 5.0)
    qa(b, torch.ones(1, device=DEV))
    f32 = np.float32
    exp_min = f32(f32(-3.0) * f32(0.95)) + f32(f32(-7.0) * f32(1 - 0.95))
    assert abs(float(qa.x_min) - float(exp_min)) < 1e-6 and abs(float(qa.x_max) - (5 * 0.95 + 1 * 0.05)) < 1e-6


def _ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("tag", ["deit_tiny_natural", "deit_small_natural", "deit_tiny_ibert_natural"])
def test_gpu_calibration_against_the_reference_trace(tag, golden_dir):
    """Row f4 pinned against the reference itself (quant_modules.py:310-360; scripts/inference.py:33-91 runs these forwards).
    tests/golden/calib_trace.npz holds, for every calibration batch and every QuantAct of the reference, the raw (min, max) it
    observed and the range it held after its update.  The module mirror runs the same batches on the GPU; after each
    observation a QuantAct is set to the reference's post-update range, so both sides quantise with identical scales and every
    later integer is the same.  Asserted:
      1. the mirror's update rule (initialise / EMA, float32) applied to the reference's observation gives the reference's
         post-update range BITWISE, for every QuantAct and batch;
      2. the mirror's own observation (HIP min / max over the float view it hands on) equals the reference's BITWISE wherever the
         observed tensor is an exact function of integers (input, LayerNorm / GELU / softmax outputs), and within 1 ulp elsewhere:
         behind a QuantLinear / QuantConv2d / QuantMatMul the reference observes `F.linear(x / s, W_int) * scale` -- a float32
         sgemm over the near-integers fl(fl(q s) / s), 0.003 of an integer step away from the exact accumulator this build
         multiplies by the same scale -- so the last bit of those extrema is the BLAS kernel's, not the
         algorithm's;
      3. I-BERT: the LayerNorms' overflow shifts equal the reference's;
      4. the final ranges (= the fixture's, by construction) drive the frozen engine to the reference's INT32 logits."""
    import json
    import ivit_amd as ivit
    from ivit_amd.checkpoint import load_synthetic_model as lsm
    tr = np.load(os.path.join(golden_dir, "calib_trace.npz"), allow_pickle=True)
    meta = json.loads(str(tr[tag + "/meta"]))
    names = [str(n) for n in tr[tag + "/names"]]
    raw_ref, post_ref = tr[tag + "/raw"], tr[tag + "/post"]          # [batch, qact, 2]
    fam = meta["family"]
    model = getattr(ivit, meta["factory"])(gelu_type=fam, softmax_type=fam, layernorm_type=fam)
    fs = synth.make_float_state(meta["factory"], meta["weight_seed"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    mods = dict(model.named_modules())
    idx = {n: i for i, n in enumerate(names)}
    assert [n for n, m in model.named_modules() if isinstance(m, q.QuantAct)] == names
    state = {"b": 0}
    raw_got = np.zeros_like(raw_ref)
    rule_bad = []

    def make_observe(name, mod):
        def observe(x_act):
            b, i = state["b"], idx[name]
            xa = x_act.detach().contiguous().float()
            mm = torch.empty(2, dtype=torch.float32, device=xa.device)
            _lib.call("ivit_minmax_f32", _lib.ptr(xa), xa.numel(), _lib.ptr(mm), _lib.stream_ptr())
            raw_got[b, i] = mm.cpu().numpy()
            # the module's own update rule on the REFERENCE's observation (its state is the reference's from the batch before)
            ref_obs = torch.from_numpy(raw_ref[b, i].copy()).to(xa.device)
            q.QuantAct._observe_update(mod, ref_obs[0], ref_obs[1])
            got = np.array([float(mod.x_min), float(mod.x_max)], np.float32)
            if not np.array_equal(got.view(np.int32), post_ref[b, i].view(np.int32)):
                rule_bad.append((b, name, got.tolist(), post_ref[b, i].tolist()))
            mod.x_min = torch.full((1,), float(post_ref[b, i, 0]), dtype=torch.float32, device=xa.device)
            mod.x_max = torch.full((1,), float(post_ref[b, i, 1]), dtype=torch.float32, device=xa.device)
        return observe

    for n in names:
        mods[n]._observe = make_observe(n, mods[n])
    with torch.no_grad():
        for b, cs in enumerate(meta["calib_seeds"]):
            state["b"] = b
            model(torch.from_numpy(synth.make_images(meta["calib_batch"], cs)).to(DEV))
    assert not rule_bad, f"update rule differs from the reference at {rule_bad[:3]} ({len(rule_bad)} cases)"
    d = np.maximum(_ulps(raw_got[..., 0], raw_ref[..., 0]), _ulps(raw_got[..., 1], raw_ref[..., 1]))     # [batch, qact]
    # QuantActs whose input is an exact function of integers: the network input, and whatever follows LayerNorm, GELU, softmax
    import re
    exact_pat = re.compile(r"^(qact_input|qact_pos|qact1|qact2|blocks\.\d+\.qact1|blocks\.\d+\.qact3|blocks\.\d+\.mlp\.qact1|"
                           r"blocks\.\d+\.attn\.int_softmax\.act)$")
    exact = [i for i, n in enumerate(names) if exact_pat.match(n)]
    assert len(exact) >= 3 * 12 + 4
    worst = [(names[i], int(d[:, i].max())) for i in np.argsort(-d.max(axis=0))[:4]]
    assert d.max() <= 1, f"observed extrema more than 1 ulp from the reference's: {worst}"
    assert all(d[:, i].max() == 0 for i in exact), [(names[i], int(d[:, i].max())) for i in exact if d[:, i].max()]
    assert (d == 0).mean() > 0.9, f"only {(d == 0).mean():.3f} of the observations are bitwise equal"
    if fam == "ibert":
        for n, sh in zip(tr[tag + "/ln_names"], tr[tag + "/ln_shift"]):
            assert float(mods[str(n)].shift) == float(sh), n
    # the calibrated model, frozen: the reference's logits
    z = lsm(tag)[4]
    ivit.freeze_model(model)
    for n in names:
        del mods[n]._observe
    m2 = json.loads(str(z["meta"]))
    imgs = torch.from_numpy(synth.make_images(m2["n_images"], m2["image_seed"])).to(DEV)
    with torch.no_grad():
        model(imgs)
    eng = model._engine[2]
    assert np.array_equal(eng.forward(imgs)[0].cpu().numpy(), z["logits_int32"])


def test_gpu_calibration_free_running_deit_tiny(golden_dir):
    """the same calibration WITHOUT forcing the ranges: every QuantAct keeps its own.  A last-bit difference of one observed
    extremum (see above) changes a scale by one ulp and with it a few roundings downstream, so agreement with the reference is not
    bitwise in general; on DeiT-T all 137 ranges land within 1 ulp (133 of them bitwise) -- recorded here as a regression bound"""
    import json
    import ivit_amd as ivit
    z = np.load(os.path.join(golden_dir, "deit_tiny_natural.npz"), allow_pickle=True)
    meta = json.loads(str(z["meta"]))
    model = getattr(ivit, meta["factory"])(gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_float_state(meta["factory"], meta["weight_seed"]).items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        for cs in meta["calib_seeds"]:
            model(torch.from_numpy(synth.make_images(meta["calib_batch"], cs)).to(DEV))
    mods = dict(model.named_modules())
    names = [str(n) for n in z["range_names"]]
    lo = np.array([float(mods[n].x_min) for n in names], np.float32)
    hi = np.array([float(mods[n].x_max) for n in names], np.float32)
    d = np.maximum(_ulps(lo, z["x_min"]), _ulps(hi, z["x_max"]))
    assert d.max() <= 1 and (d == 0).sum() >= 130, (int(d.max()), int((d == 0).sum()))


def test_checkpoint_harness_end_to_end(tmp_path):
    """save a reference-format checkpoint of the synthetic DeiT-S, load it back strictly, evaluate on the golden images
    with the golden top-1 as labels (scripts/inference.py load_model -> evaluate_dataset)"""
    from ivit_amd import inference
    model, meta, z = load_model("deit_small")
    path = tmp_path / "ckpt.pth.tar"
    inference.save_checkpoint(model, path, {"model_name": "deit_small_patch16_224", "gelu_type": "ivit",
                                            "softmax_type": "ivit", "layernorm_type": "ivit"}, epoch=0)
    m2 = inference.load_model(path, device=DEV, strict_load=True)
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"]))
    labels = torch.from_numpy(z["top1"])
    loader = [(imgs[:2], labels[:2]), (imgs[2:], labels[2:])]
    assert inference.evaluate_dataset(m2, loader, DEV, print_batch_stats=False) == (100.0, 100.0, 100.0)
    wrong = [(imgs, (labels + 1) % 1000)]
    t1, _, _ = inference.evaluate_dataset(m2, wrong, DEV, print_batch_stats=False)
    assert t1 == 0.0
    with torch.no_grad():
        assert np.array_equal(bits(m2(imgs.to(DEV))), z["logits_f32_bits"])
    # non-strict load with the random warm-up forward, then frozen (inference.py:209-223)
    m3 = inference.load_model(path, device=DEV, strict_load=False, use_random_calibration_warmup=True)
    assert m3.is_frozen()


# ----------------------------------------------------------------------------------- I-BERT operator family (row f3)
@pytest.fixture(scope="module")
def ikat(golden_dir):
    return np.load(os.path.join(golden_dir, "ibert_kat.npz"))


def test_ibert_gelu_module_kat(ikat):
    g = q.IBERTIntGELU().to(DEV)
    for i in range(5):
        k, s = ikat[f"gelu{i}_k"].astype(np.float32), ikat[f"gelu{i}_s"]
        x = torch.from_numpy((k * s).astype(np.float32)).to(DEV)
        y, so = g(x, torch.tensor([float(s)], device=DEV))
        assert float(so) == float(ikat[f"gelu{i}_sout"])
        want = (ikat[f"gelu{i}_out"].astype(np.float32) * ikat[f"gelu{i}_sout"]).astype(np.float32)
        assert np.array_equal(bits(y), want.view(np.int32))


def test_ibert_softmax_module_kat(ikat):
    sm = q.IBERTIntSoftmax(8).to(DEV)
    for i in range(5):
        k, s = ikat[f"softmax{i}_k"].astype(np.float32), ikat[f"softmax{i}_s"]
        lo, hi = ikat[f"softmax{i}_range"]
        sm.act.x_min.fill_(float(lo))
        sm.act.x_max.fill_(float(hi))
        sm.act.fix()
        x = torch.from_numpy((k * s).astype(np.float32)).to(DEV)
        y, so = sm(x, torch.tensor([float(s)], device=DEV))
        assert float(so) == 2.0 ** -7
        want = (ikat[f"softmax{i}_out"].astype(np.float32) * np.float32(2.0 ** -7)).astype(np.float32)
        assert np.array_equal(bits(y), want.view(np.int32)), i
    # calibration mode: the internal QuantAct(16) initialises its range from exp_int of the batch
    sm2 = q.IBERTIntSoftmax(8).to(DEV)
    k, s = ikat["softmax0_k"].astype(np.float32), ikat["softmax0_s"]
    sm2(torch.from_numpy((k * s).astype(np.float32)).to(DEV), torch.tensor([float(s)], device=DEV))
    from oracle import ibert as ib
    ex = ib.softmax(k, s, -1.0, 1.0, return_exp=True)[3]
    assert float(sm2.act.x_min) == float(ex.min()) and float(sm2.act.x_max) == float(ex.max())


def test_ibert_layernorm_module_kat(ikat):
    for i in range(4):
        k, s = ikat[f"ln{i}_k"].astype(np.float32), ikat[f"ln{i}_s"]
        C = k.shape[-1]
        ln = q.IBERTIntLayerNorm(C).to(DEV)
        ln.weight.data = torch.from_numpy(ikat[f"ln{i}_gamma"]).to(DEV)
        ln.bias.data = torch.from_numpy(ikat[f"ln{i}_beta"]).to(DEV)
        ln.fix()
        y, so = ln(torch.from_numpy((k * s).astype(np.float32)).to(DEV), torch.tensor([float(s)], device=DEV))
        assert np.array_equal(so.cpu().numpy(), ikat[f"ln{i}_sout"])
        got, want = y.cpu().numpy(), ikat[f"ln{i}_out_bits"].view(np.float32)
        nan = np.isnan(want)                       # the constant row: std = 0 -> 0 * inf (NaN payload / sign is not compared)
        assert np.array_equal(np.isnan(got), nan) and nan.sum() == C
        assert np.array_equal(got[~nan].view(np.int32), want[~nan].view(np.int32))


@pytest.mark.parametrize("C,s_in,shift", [(192, 2.0 ** -4, 0), (768, 0.0371, 0), (384, 0.05, 0), (1024, 0.0213, 0), (100, 0.0371, 0),
                                          (198, 0.0371, 0), (768, 0.0371, 2), (384, 2.0 ** -5, 1), (198, 0.05, 1)])
def test_ibert_layernorm_i8_engine_kernel_equals_modules(C, s_in, shift):
    """ivit_ibert_layernorm_i8 (the fused engine's kernel: sums decided from exact integers, undecided rows literally) ==
    IBERTIntLayerNorm (literal float32 kernel) followed by a QuantAct, on random rows, rows whose mean is an exact tie
    (sum q = C * k + C / 2: the float32 reduction order decides), constant rows + one outlier, saturating rows"""
    rng = np.random.default_rng(C)
    rows = 257
    qv = rng.integers(-128, 128, size=(rows, C)).astype(np.int8)
    qv[0] = 5
    qv[0, 3] = 90                                        # tiny variance
    for r in range(1, 40):                               # exact mean ties
        row = rng.integers(-100, 100, size=C).astype(np.int64)
        want = C * int(rng.integers(-20, 20)) + C // 2
        row[0] += want - int(row.sum())
        while abs(row[0]) > 127:                         # spread the correction
            i = int(rng.integers(1, C))
            step = int(np.clip(row[0], -60, 60))
            if -128 <= row[i] + step <= 127:
                row[i] += step
                row[0] -= step
        assert row.sum() == want and np.abs(row).max() <= 128
        qv[r] = np.clip(row, -128, 127)
    qv[40] = 127
    qv[40, ::2] = -128
    gamma = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    beta = rng.uniform(-1, 1, size=C).astype(np.float32)
    ln = q.IBERTIntLayerNorm(C).to(DEV)
    ln.weight.data = torch.from_numpy(gamma).to(DEV)
    ln.bias.data = torch.from_numpy(beta).to(DEV)
    ln.shift.fill_(float(shift))          # the overflow buffer a calibration pass may have raised (ibert_modules.py:134-137)
    ln.fix()
    act = q.QuantAct().to(DEV)
    act.x_min.fill_(-2.9)
    act.x_max.fill_(3.1)
    act.fix()
    s_t = torch.tensor([s_in], dtype=torch.float32, device=DEV)
    x = (torch.from_numpy(qv.astype(np.float32)).to(DEV) * s_t)
    with torch.no_grad():
        y, s_ln = ln(x, s_t)
        z, s_z = act(y, s_ln)
    exp = torch.round(z / s_z).to(torch.int32).cpu().numpy()
    lp = LayerNormParams(gamma, beta, float(s_z))
    out = torch.zeros(rows, C, dtype=torch.int8, device=DEV)
    dq, db, dsl = torch.from_numpy(qv).to(DEV), torch.from_numpy(lp.bias_int).to(DEV), torch.from_numpy(lp.s_ln).to(DEV)
    dm, de = torch.from_numpy(lp.m.view(np.int32)).to(DEV), torch.from_numpy(lp.e).to(DEV)
    _lib.call("ivit_ibert_layernorm_i8", _lib.ptr(dq), C, rows, C, float(s_in), _lib.ptr(db), _lib.ptr(dsl), float(2.0 ** shift), _lib.ptr(dm),
              _lib.ptr(de), _lib.ptr(out), C, 0, _lib.stream_ptr())
    got = out.cpu().numpy().astype(np.int32)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, (len(bad), bad[:5], got[tuple(bad[0])], exp[tuple(bad[0])])
    assert np.abs(exp).max() > 50


@pytest.mark.parametrize("C,s_in,shift", [(384, 2.0 ** -9, 0), (768, 1.37e-4, 1), (198, 3.1e-4, 0)])
def test_ibert_layernorm_i16_kernel_equals_modules(C, s_in, shift):
    """ivit_ibert_layernorm_i16_i8 (the 16-bit residual stream's LayerNorm under the I-BERT operators) == IBERTIntLayerNorm + QuantAct
    on int16 rows: random, tiny-variance, full-range"""
    rng = np.random.default_rng(C + 1)
    rows = 131
    qv = np.clip(np.rint(rng.normal(0, 6000, size=(rows, C))), -32768, 32767).astype(np.int16)
    qv[0] = 7
    qv[0, 3] = 9000
    qv[1] = 32767
    qv[1, ::2] = -32768
    qv[2] = rng.integers(-40, 40, size=C)
    gamma = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    beta = rng.uniform(-1, 1, size=C).astype(np.float32)
    ln = q.IBERTIntLayerNorm(C).to(DEV)
    ln.weight.data = torch.from_numpy(gamma).to(DEV)
    ln.bias.data = torch.from_numpy(beta).to(DEV)
    ln.shift.fill_(float(shift))
    ln.fix()
    act = q.QuantAct().to(DEV)
    act.x_min.fill_(-2.9)
    act.x_max.fill_(3.1)
    act.fix()
    s_t = torch.tensor([s_in], dtype=torch.float32, device=DEV)
    x = (torch.from_numpy(qv.astype(np.float32)).to(DEV) * s_t)
    with torch.no_grad():
        y, s_ln = ln(x, s_t)
        z, s_z = act(y, s_ln)
    exp = torch.round(z / s_z).to(torch.int32).cpu().numpy()
    lp = LayerNormParams(gamma, beta, float(s_z))
    out = torch.zeros(rows, C, dtype=torch.int8, device=DEV)
    dq, db, dsl = torch.from_numpy(qv).to(DEV), torch.from_numpy(lp.bias_int).to(DEV), torch.from_numpy(lp.s_ln).to(DEV)
    dm, de = torch.from_numpy(lp.m.view(np.int32)).to(DEV), torch.from_numpy(lp.e).to(DEV)
    _lib.call("ivit_ibert_layernorm_i16_i8", _lib.ptr(dq), C, rows, C, float(s_in), _lib.ptr(db), _lib.ptr(dsl), float(2.0 ** shift),
              _lib.ptr(dm), _lib.ptr(de), _lib.ptr(out), C, _lib.stream_ptr())
    got = out.cpu().numpy().astype(np.int32)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, (len(bad), bad[:5], got[tuple(bad[0])], exp[tuple(bad[0])])
    from ivit_amd.prepare import markstein_division_ok
    if markstein_division_ok(s_in, 16):       # the three-instruction quotient, where the host check allows it
        out2 = torch.zeros(rows, C, dtype=torch.int8, device=DEV)
        _lib.call("ivit_ibert_layernorm_i16_i8_ex", _lib.ptr(dq), C, rows, C, float(s_in), _lib.ptr(db), _lib.ptr(dsl), float(2.0 ** shift),
                  _lib.ptr(dm), _lib.ptr(de), _lib.ptr(out2), C, 1, _lib.stream_ptr())
        assert torch.equal(out2, out)
    assert np.abs(exp).max() > 50


def test_ibert_model_module_path_matches_reference_golden():
    """DeiT-T with gelu / softmax / layernorm = 'ibert' (the fork's default operator family) through the module path:
    float logits bitwise equal to the reference's."""
    model, meta, z = load_model("deit_tiny_ibert")
    assert meta["family"] == "ibert" and type(model.blocks[0].attn.int_softmax).__name__ == "IBERTIntSoftmax"
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    assert model.engine_unsupported_reason() is None and model.takes_engine(imgs)
    with torch.no_grad():
        ye = model(imgs)            # default: the fused engine (family "ibert")
        model.use_engine = False
        y = model(imgs)             # module by module
    assert np.array_equal(bits(y), z["logits_f32_bits"][:2])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"][:2])
    assert np.array_equal(bits(ye), bits(y))


def test_mixed_operator_families_take_the_module_path():
    """the fused engine implements all-'ivit' and all-'ibert'; a mixture (the reference's registry allows any combination,
    vit_quant.py:188-190) runs module by module and says why"""
    fs = synth.make_float_state("deit_tiny_patch16_224", 5)
    model = ivit.deit_tiny_patch16_224(gelu_type="ibert", softmax_type="ivit", layernorm_type="ivit")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    imgs = torch.from_numpy(synth.make_images(2, 3)).to(DEV)
    with torch.no_grad():
        model(imgs)
    ivit.freeze_model(model)
    assert "operator family" in model.engine_unsupported_reason() and not model.takes_engine(imgs)
    with torch.no_grad():
        y = model(imgs)
    assert y.shape == (2, 1000) and torch.isfinite(y).all()


@pytest.mark.parametrize("regime,size", [("pow2", "small"), ("natural", "small"), ("natural", "base")])
def test_ibert_engine_equals_module_path_deit_small_width(regime, size):
    """wider I-BERT models than the reference fixtures cover (DeiT-S: C = 384, 6 heads; DeiT-B: C = 768, 12 heads; 12 fresh
    images = 2364 token rows: the large-batch kernels incl. the GELU map in the fc1 epilogue; ranges calibrated here, snapped to
    powers of two or left as calibrated): fused engine == module-by-module path, float logits bitwise"""
    fs = synth.make_float_state(f"deit_{size}_patch16_224", 23)
    model = getattr(ivit, f"deit_{size}_patch16_224")(gelu_type="ibert", softmax_type="ibert", layernorm_type="ibert")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(4, 77)).to(DEV))         # calibration forward (running min / max)
    if regime == "pow2":
        for mod in model.modules():
            if isinstance(mod, q.QuantAct):
                qmax = 2 ** (mod.activation_bit - 1) - 1
                a = max(-float(mod.x_min), float(mod.x_max)) / qmax
                p = 2.0 ** np.ceil(np.log2(a))
                mod.x_max.fill_(qmax * p)
                mod.x_min.fill_(-qmax * p)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(12, 78)).to(DEV)
    assert model.takes_engine(imgs), model.engine_unsupported_reason()
    with torch.no_grad():
        ye = model(imgs)
        assert model.engine(12).family == "ibert"
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(bits(ye), bits(ym))
    assert len(set(ym.argmax(dim=1).cpu().tolist())) > 1


# ----------------------------------------------------------------------------------- engine dispatch (dispatch.py)
@pytest.mark.parametrize("tag", ["deit_tiny_w16", "deit_tiny_w16all"])
def test_non_8bit_widths_match_the_reference(tag):
    """the reference's width knobs (vit_quant.py:180-187): a DeiT-T with a 16-bit residual stream -- and, `w16all`, what
    `--bitwidth 16` sets (quant_train.py:299-306): every knob at 16 incl. the 16-bit Shiftmax output feeding P.V and the
    16-bit position embedding -- is NOT what the fused int8 engine computes: the mirror must say so and run it module by
    module, reproducing the reference's logits"""
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    model = ivit.deit_tiny_patch16_224(**meta["widths"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct):
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    # the 16-bit residual stream -- with the softmax output and the position embedding at 8 bits (w16) or at 16 (w16all) -- is a
    # pattern the fused engine implements (IntViTEngine(stream_bits=16, ...)): the default path, and it reproduces the reference's
    # logits, too
    assert model.engine_unsupported_reason() is None and model.takes_engine(imgs)
    with torch.no_grad():
        ye = model(imgs)
    eng = model.engine(2)
    assert (eng.stream_bits, eng.softmax_bits, eng.pos_bits) == ((16, 8, 8) if tag == "deit_tiny_w16" else (16, 16, 16))
    assert np.array_equal(bits(ye), z["logits_f32_bits"][:2])
    model.use_engine = False
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"][:2])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"][:2])
    # and it really is a different function from the all-8-bit model with the same weights
    z8 = load_synthetic_model("deit_tiny")[4]
    assert not np.array_equal(z["logits_f32_bits"][:2], z8["logits_f32_bits"][:2])


def test_ibert_int16_matches_the_reference():
    """the authors' INT16 configuration with the fork's default operators (I-BERT; every width knob at 16, ranges as calibrated,
    LayerNorm overflow shifts 1 and 2 from the calibration pass): fixture deit_tiny_ibert_w16all.npz from the reference.  The
    module path reproduces every QuantAct tap, the INT32 logits and top-1 -- incl. the 16-bit IBERTIntSoftmax output (p up to
    2^15) through QuantMatMul -- and the fused engine (stream / softmax / pos_embed at 16 bits) gives the module path's float
    logits bitwise"""
    import zlib
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny_ibert_w16all")
    assert meta["family"] == "ibert" and set(meta["widths"].values()) == {16}
    model = ivit.deit_tiny_patch16_224(gelu_type="ibert", softmax_type="ibert", layernorm_type="ibert", **meta["widths"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    mods = dict(model.named_modules())
    for name, bw in zip([str(n) for n in z["range_names"]], z["range_bits"]):
        assert int(mods[name].activation_bit) == int(bw), name
        mods[name].x_min.fill_(float(ranges[name][0]))
        mods[name].x_max.fill_(float(ranges[name][1]))
    assert sorted(set(meta["ln_shifts"].values())) == [1.0, 2.0]
    for name, sh in meta["ln_shifts"].items():
        mods[name].shift.fill_(float(sh))
    model.to(DEV)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    assert model.takes_engine(imgs), model.engine_unsupported_reason()
    with torch.no_grad():
        ye = model(imgs).clone()
    eng = model.engine(2)
    assert (eng.family, eng.stream_bits, eng.softmax_bits, eng.pos_bits) == ("ibert", 16, 16, 16)
    model.use_engine = False
    got = {}

    def hook(name):
        def fn(mod, inp, outp):
            y, sc = outp
            got[name] = zlib.crc32(np.ascontiguousarray(torch.round(y / sc).to(torch.int64).cpu().numpy().astype(np.int32)).tobytes())
        return fn

    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct) and not name.endswith("int_softmax.act"):
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        y = model(imgs)
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    bad = [n for n in got if n in gold and got[n] != int(gold[n])]
    assert len(got) >= 100 and not bad, bad[:6]
    li = np.rint(y.cpu().numpy().astype(np.float64) / z["head_scale"].astype(np.float64)).astype(np.int32)
    assert np.array_equal(li, z["logits_int32"])
    # (natural scales: the reference's head multiplies fl(fl(q s) / s) in a float32 GEMM -- its float logits carry that noise in
    # the last bits; the contract is the INT32 logits)
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"])
    assert np.array_equal(bits(ye), bits(y))


def test_io_stats_collector(tmp_path):
    """attach_io_stat_hooks / save_io_stats_df (quant_modules.py:17-125; what scripts/inference.py:367-400 does by default): one
    record per sub-module call with the extrema in float and integer units; the model runs module by module while the collector is
    on (its hooks sit on the sub-modules) and goes back to the fused engine when it is disabled"""
    model, meta, z = load_model("deit_tiny")
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    ivit.clear_io_stats()
    ivit.enable_io_stats()
    ivit.attach_io_stat_hooks(model)
    assert not model.takes_engine(imgs)
    cap = {}
    model.blocks[3].qact2.register_forward_hook(lambda m, i, o: cap.update(y=o[0].detach().clone(), s=o[1].detach().clone()))
    with torch.no_grad():
        y = model(imgs)
    assert np.array_equal(bits(y), z["logits_f32_bits"][:2])
    df = ivit.save_io_stats_df(str(tmp_path / "io_stats_val.pkl"), to_csv=True)
    assert (tmp_path / "io_stats_val.pkl").exists() and (tmp_path / "io_stats_val.csv").exists()
    assert {"layer", "type", "min_in", "max_in", "min_out", "max_out", "scale_in", "scale_out", "min_in_int", "max_in_int",
            "min_out_int", "max_out_int", "shape_in", "shape_out", "min_A_int", "max_B_int", "shape_A"} <= set(df.columns)
    row = df[df.layer == "blocks.3.qact2"].iloc[0]
    yi = cap["y"] / cap["s"]
    assert row["type"] == "QuantAct" and row["min_out_int"] == yi.min().item() and row["max_out_int"] == yi.max().item()
    assert row["min_out"] == cap["y"].min().item() and row["scale_out"] == cap["s"].item() and tuple(row["shape_out"]) == (2, 197, 192)
    assert -128 <= row["min_out_int"] < 0 < row["max_out_int"] <= 127
    mm = df[df.type == "QuantMatMul"]
    assert len(mm) == 24 and (mm["max_B_int"] <= 127).all() and (mm["min_A_int"] >= -128).all()
    # the reference's quirk (quant_modules.py:79-81): a layer whose scale is per channel raises inside the hook and
    # is silently dropped -- every QuantLinear, every LayerNorm and the QuantAct behind them
    layers = set(df.layer)
    assert (df.type == "QuantAct").sum() >= 60 and "blocks.11.qact4" in layers and "blocks.0.attn.qkv" not in layers and "blocks.11.norm2" not in layers
    n = len(df)
    ivit.disable_io_stats()
    assert model.takes_engine(imgs)
    with torch.no_grad():
        y2 = model(imgs)
    assert np.array_equal(bits(y2), z["logits_f32_bits"][:2]) and len(ivit.get_io_stats_df()) == n
    ivit.clear_io_stats()
    ivit.enable_io_stats()
    assert len(ivit.get_io_stats_df()) == 0


def test_unsupported_width_pattern_takes_the_module_path():
    """only att_block_out_bw = 16 (one of the reference's sweep points): not a pattern of the fused engine -> module path"""
    fs = synth.make_float_state("deit_tiny_patch16_224", 5)
    model = ivit.deit_tiny_patch16_224(att_block_out_bw=16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    imgs = torch.from_numpy(synth.make_images(2, 3)).to(DEV)
    with torch.no_grad():
        model(imgs)
    ivit.freeze_model(model)
    assert "16-bit" in model.engine_unsupported_reason() and not model.takes_engine(imgs)
    with torch.no_grad():
        y = model(imgs)
    assert model._engine is None and torch.isfinite(y).all()


def test_ibert_int16_engine_equals_module_path():
    """the authors' 'I-BERT INT16' configuration (.vscode/launch.json: --bitwidth 16 with the I-BERT operators): DeiT-S, every
    width knob at 16, ranges as calibrated, 12 fresh images: fused engine (stream_bits = softmax_bits = pos_bits = 16,
    family 'ibert') == module-by-module path, float logits bitwise"""
    w = {k: 16 for k in ("patch_embed_bw", "pos_encoding_bw", "block_input_bw", "attention_out_bw", "softmax_bw", "mlp_out_bw",
                         "norm2_in_bw", "att_block_out_bw")}
    fs = synth.make_float_state("deit_small_patch16_224", 41)
    model = ivit.deit_small_patch16_224(gelu_type="ibert", softmax_type="ibert", layernorm_type="ibert", **w)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(4, 93)).to(DEV))
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(12, 94)).to(DEV)
    assert model.takes_engine(imgs), model.engine_unsupported_reason()
    with torch.no_grad():
        ye = model(imgs)
        eng = model.engine(12)
        assert (eng.family, eng.stream_bits, eng.softmax_bits, eng.pos_bits) == ("ibert", 16, 16, 16)
        eng.fuse_res16 = eng.fuse_ibert_gelu = False      # the unfused kernels give the same integers
        yu = model(imgs)
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(bits(ye), bits(ym)) and np.array_equal(bits(yu), bits(ym))
    assert len(set(ym.argmax(dim=1).cpu().tolist())) > 1


@pytest.mark.parametrize("family,pattern", [("ivit", "w16"), ("ivit", "all16"), ("ibert", "all16")])
def test_16bit_engine_fused_kernels_at_deit_base_width(family, pattern):
    """DeiT-B width (C = 768: the weights-in-registers GEMMs, i.e. projection / fc2 + 16-bit QuantAct + residual QuantAct in one
    kernel, in place on the int16 stream), 3 blocks, 13 images (2561 token rows: 20 full tiles + a ragged one): engine == its
    unfused form == the module-by-module path, float logits bitwise"""
    from ivit_amd.vit_quant import VisionTransformer
    a16 = 16 if pattern == "all16" else 8
    w = dict(patch_embed_bw=16, pos_encoding_bw=a16, block_input_bw=16, attention_out_bw=16, softmax_bw=a16, mlp_out_bw=16,
             norm2_in_bw=16, att_block_out_bw=16)
    fs = synth.make_float_state("deit_base_patch16_224", 51, depth=3)
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=3, num_heads=12, mlp_ratio=4, qkv_bias=True, gelu_type=family,
                              softmax_type=family, layernorm_type=family, **w)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(4, 95)).to(DEV))
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(13, 96)).to(DEV)
    assert model.takes_engine(imgs), model.engine_unsupported_reason()
    with torch.no_grad():
        ye = model(imgs).clone()
        eng = model.engine(13)
        assert eng.stream_bits == 16 and eng.family == family
        assert all(b[k].get("Wf") is not None for b in eng.blocks for k in ("proj", "fc2", "fc1"))      # the fused kernels ran
        eng.fuse_res16 = eng.fuse_ibert_gelu = False
        yu = model(imgs).clone()
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(bits(ye), bits(ym)) and np.array_equal(bits(yu), bits(ym))
    assert len(set(ym.argmax(dim=1).cpu().tolist())) > 1


@pytest.mark.parametrize("regime", ["pow2", "natural"])
def test_16bit_stream_engine_equals_module_path_deit_small(regime):
    """DeiT-S with the 16-bit residual stream (patch_embed / block_input / attention_out / mlp_out / norm2_in / att_block_out at
    16 bits), 12 fresh images (2364 token rows: persistent GEMMs, int16 LayerNorm, 16-bit residual kernels): engine ==
    module-by-module path, float logits bitwise, with power-of-two ranges and with ranges as calibrated"""
    w = dict(patch_embed_bw=16, pos_encoding_bw=8, block_input_bw=16, attention_out_bw=16, softmax_bw=8, mlp_out_bw=16, norm2_in_bw=16,
             att_block_out_bw=16)
    if regime == "natural":     # ... and here every knob at 16 ('--bitwidth 16'): 16-bit Shiftmax output into P.V, 16-bit pos_embed
        w.update(pos_encoding_bw=16, softmax_bw=16)
    fs = synth.make_float_state("deit_small_patch16_224", 31)
    model = ivit.deit_small_patch16_224(**w)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        model(torch.from_numpy(synth.make_images(4, 91)).to(DEV))
    if regime == "pow2":
        for mod in model.modules():
            if isinstance(mod, q.QuantAct):
                qmax = 2 ** (mod.activation_bit - 1) - 1
                a = max(-float(mod.x_min), float(mod.x_max)) / qmax
                p = 2.0 ** np.ceil(np.log2(a))
                mod.x_max.fill_(qmax * p)
                mod.x_min.fill_(-qmax * p)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(12, 92)).to(DEV)
    assert model.takes_engine(imgs), model.engine_unsupported_reason()
    with torch.no_grad():
        ye = model(imgs)
        assert model.engine(12).stream_bits == 16
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(bits(ye), bits(ym))
    assert len(set(ym.argmax(dim=1).cpu().tolist())) > 1


def test_engine_follows_load_state_dict_and_range_changes():
    """the cached engine is a snapshot: new weights, new ranges or a re-freeze must rebuild it"""
    model, meta, z = load_model("deit_tiny")
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y0 = model(imgs).clone()
        assert np.array_equal(bits(y0), z["logits_f32_bits"][:2])
        e0 = model._engine[2]
        assert model(imgs[:1]).shape == (1, 1000) and model._engine[2] is e0          # smaller batch: same engine
        # 1. new weights through load_state_dict
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        sd["head.weight"] = sd["head.weight"].flip(0)
        sd["head.bias"] = sd["head.bias"].flip(0)
        model.load_state_dict(sd)
        y1 = model(imgs)
        assert model._engine[2] is not e0
        assert not torch.equal(y1, y0)
        model.use_engine = False
        assert np.array_equal(bits(model(imgs)), bits(y1))       # engine == module path on the new weights
        model.use_engine = True
        # 2. an in-place parameter edit (optimizer-style)
        e1 = model._engine[2]
        model.blocks[0].mlp.fc2.bias.mul_(2.0).add_(0.05)
        y2 = model(imgs)
        assert model._engine[2] is not e1 and not torch.equal(y2, y1)
        # 3. unfreeze -> recalibrate on other data -> freeze
        e2 = model._engine[2]
        ivit.unfreeze_model(model)
        model(torch.from_numpy(synth.make_images(2, 99)).to(DEV) * 3.0)
        ivit.freeze_model(model)
        y3 = model(imgs)
        assert model._engine[2] is not e2 and not torch.equal(y3, y2)
        model.use_engine = False
        assert np.array_equal(bits(model(imgs)), bits(y3))


@pytest.mark.parametrize("num_classes", [100, 10, 37])
def test_engine_any_class_count(num_classes):
    """head width comes from the model, not from a constant: engine == module path for num_classes != 1000"""
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny")
    model = ivit.deit_tiny_patch16_224(num_classes=num_classes)
    sd = {k: torch.from_numpy(v) for k, v in fs.items()}
    sd["head.weight"] = sd["head.weight"][:num_classes].clone()
    sd["head.bias"] = sd["head.bias"][:num_classes].clone()
    model.load_state_dict(sd, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct):
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    imgs = torch.from_numpy(synth.make_images(3, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        ye = model(imgs)
        assert model._engine is not None and ye.shape == (3, num_classes)
        model.use_engine = False
        ym = model(imgs)
    assert np.array_equal(bits(ye), bits(ym))
    # the per-class scale differs per row of the head, so logits of the kept classes equal the 1000-class model's
    assert np.array_equal(bits(ye), z["logits_f32_bits"][:3, :num_classes])


def test_ibert_natural_scale_module_path_matches_reference():
    """the fork's default operator family with ranges AS CALIBRATED (the regime of its authors' checkpoints): the I-BERT
    modules run the literal float kernels (x / s itself, torch's reduction order) and reproduce the reference's INT32 logits,
    top-1 and every QuantAct tap"""
    import zlib
    model, meta, z = load_model("deit_tiny_ibert_natural")
    assert meta["family"] == "ibert" and meta["regime"] == "natural"
    model.use_engine = False          # (the fused engine on this fixture: test_gpu_model.py)
    got = {}

    def hook(name):
        def fn(mod, inp, outp):
            y, s = outp
            got[name] = zlib.crc32(np.ascontiguousarray(torch.round(y / s).to(torch.int64).cpu().numpy().astype(np.int32)).tobytes())
        return fn

    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct) and not name.endswith("int_softmax.act"):
            mod.register_forward_hook(hook(name))
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y = model(imgs)
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    bad = [n for n in got if n in gold and got[n] != int(gold[n])]
    assert not bad, bad[:6]
    li = np.rint(y.cpu().numpy().astype(np.float64) / z["head_scale"].astype(np.float64)).astype(np.int32)
    assert np.array_equal(li, z["logits_int32"])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"])


# ----------------------------------------------------------------------------------- int8-carrying module path (lazy.py)
@pytest.mark.parametrize("tag", ["deit_tiny", "deit_tiny_natural", "deit_small_natural", "deit_base", "vit_base", "deit_tiny_ibert",
                                 "deit_tiny_ibert_natural", "vit_large", "vit_large_natural"])
def test_frozen_module_path_carries_int8_and_never_syncs(tag):
    """The reference's call protocol (vit_quant.py:61-90, 142-155, 285-312: QuantLinear, QuantAct, IVITIntLayerNorm, ... one by
    one) on a frozen model, with every device -> host read-back an ERROR (torch.cuda.set_sync_debug_mode): int8 payloads move
    between the modules, each QuantAct launches one fused integer kernel, all (m, e) pairs and tables come from host-side scales
    cached at the warm-up forward.  Float logits bitwise and every QuantAct tap of the reference's goldens."""
    import zlib
    from ivit_amd.quantization_utils import lazy
    model, meta, z = load_model(tag)
    model.use_engine = False
    n = min(meta["n_images"], 4)
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])[:n]).to(DEV)
    with torch.no_grad():
        model(imgs)                      # warm-up: integer weights, (m, e) pairs and tables are derived and cached
    taps = {}

    def hook(name):
        def fn(mod, inp, outp):
            y = outp[0]
            if isinstance(y, lazy.QT) and y.q8 is not None:
                taps[name] = y.q8.clone()
        return fn

    lazy.STATS.update(fused=0, materialised=0)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        with torch.no_grad():
            y = model(imgs)
            stats = dict(lazy.STATS)
            handles = [mod.register_forward_hook(hook(name)) for name, mod in model.named_modules() if isinstance(mod, q.QuantAct)]
            y2 = model(imgs)             # the same with a hook looking at every QuantAct's payload
    finally:
        torch.cuda.set_sync_debug_mode("default")
    for h in handles:
        h.remove()
    lazy.STATS.update(stats)
    assert torch.equal(y, y2)
    depth = len(model.blocks)
    # per block: LN, qkv, attention, proj, residual, LN, fc1, GELU, fc2, residual; stem: patch GEMM, cls/pos assembly; tail: LN
    assert lazy.STATS["fused"] == 10 * depth + 3, lazy.STATS
    assert lazy.STATS["materialised"] == 1, lazy.STATS          # the logits, at the model's boundary
    if "regime" not in meta:
        assert np.array_equal(bits(y), z["logits_f32_bits"][:n])
    else:     # ranges as calibrated: the reference's float logits come out of a float GEMM; its INT32 logits are the contract
        from ivit_amd.prepare import LinearParams, sym_scale
        hs = LinearParams(model.head.weight.detach().cpu().numpy(), model.head.bias.detach().cpu().numpy(),
                          sym_scale(float(model.qact2.x_min), float(model.qact2.x_max))).s_acc
        li = np.rint(y.cpu().numpy().astype(np.float64) / hs.astype(np.float64)).astype(np.int32)
        assert np.array_equal(li, z["logits_int32"][:n])
    assert np.array_equal(y.argmax(dim=1).cpu().numpy(), z["top1"][:n])
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    if n == meta["n_images"]:
        checked = [name for name in taps if name in gold]
        bad = [name for name in checked
               if zlib.crc32(np.ascontiguousarray(taps[name].cpu().numpy().astype(np.int32)).tobytes()) != int(gold[name])]
        assert not bad, bad[:6]
        assert len(checked) >= 7 * depth + 3


def test_int8_carrying_path_follows_an_in_place_weight_edit():
    """A frozen model whose weights are edited IN PLACE while every QuantAct range stays byte-identical: the integer weights, the
    accumulator scales AND the per-channel (m, e) multipliers of the fused GEMMs must all be rebuilt (round-3 advisor finding: the
    multipliers were cached without the weight version).  The int8-carrying path against the float module path (IVIT_LAZY=0)."""
    import warnings
    from ivit_amd.quantization_utils import lazy
    model, meta, z = load_model("deit_tiny")
    model.use_engine = False
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    with torch.no_grad():
        y_before = model(imgs)
        g = torch.Generator(device="cpu").manual_seed(5)
        for lin in (model.blocks[2].attn.proj, model.blocks[5].mlp.fc1, model.blocks[7].attn.qkv):
            # per-channel rescaling changes fc_scaling_factor per channel, i.e. every (m, e); ranges are untouched (frozen)
            scale = (0.5 + torch.rand(lin.weight.shape[0], 1, generator=g)).to(DEV)
            lin.weight.mul_(scale)
            lin.bias.mul_(scale.reshape(-1))
        y_lazy = model(imgs)
        old = lazy.ENABLED
        try:
            lazy.ENABLED = False
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                y_float = model(imgs)
        finally:
            lazy.ENABLED = old
    assert not torch.equal(y_before, y_lazy), "the edit changed nothing: the test has no teeth"
    assert np.array_equal(bits(y_lazy), bits(y_float)), "stale constants on the int8-carrying path after an in-place weight edit"


def test_int8_carrying_path_falls_back_to_floats_where_a_caller_looks():
    """anything that is not the model's own call protocol sees the float tensor the reference's module returns: a forward hook
    doing arithmetic on a QuantAct output, an unfrozen QuantAct in the middle of the model, IVIT_LAZY=0"""
    from ivit_amd.quantization_utils import lazy
    model, meta, z = load_model("deit_tiny")
    model.use_engine = False
    imgs = torch.from_numpy(synth.make_images(2, meta["image_seed"])).to(DEV)
    seen = {}

    def hook(mod, inp, outp):
        y, s = outp
        seen["q"] = torch.round(y / s).to(torch.int32)        # plain torch arithmetic on the output
        seen["type"] = type(y)
    h = model.blocks[3].qact2.register_forward_hook(hook)
    lazy._WARNED.clear()
    with torch.no_grad(), pytest.warns(RuntimeWarning, match="materialised as float32 inside the model"):
        y = model(imgs)          # one warning per kind of producer: the caller is told that this forward left the int8 path
    h.remove()
    assert seen["type"] is lazy.QT and int(seen["q"].abs().max()) <= 128
    assert np.array_equal(bits(y), z["logits_f32_bits"][:2])
    old = lazy.ENABLED
    try:
        lazy.ENABLED = False
        with torch.no_grad():
            y0 = model(imgs)
    finally:
        lazy.ENABLED = old
    assert np.array_equal(bits(y0), z["logits_f32_bits"][:2])
    # softmax / GELU / matmul outputs read as floats from inside the pending chain
    got = {}
    hs = [model.blocks[0].attn.int_softmax.register_forward_hook(lambda m, i, o: got.__setitem__("p", (o[0] / o[1]).amax())),
          model.blocks[0].mlp.act.register_forward_hook(lambda m, i, o: got.__setitem__("g", (o[0] * 1.0).abs().amax())),
          model.blocks[0].attn.matmul_2.register_forward_hook(lambda m, i, o: got.__setitem__("pv", o[0].float().abs().amax()))]
    with torch.no_grad():
        y1 = model(imgs)
    for h in hs:
        h.remove()
    assert 0 < float(got["p"]) <= 128 and float(got["g"]) > 0 and float(got["pv"]) > 0
    assert np.array_equal(bits(y1), z["logits_f32_bits"][:2])


@pytest.mark.parametrize("tag", ["deit_base", "deit_base_natural"])
def test_int8_module_path_fuses_the_residual_gemm_at_large_batch(tag):
    """DeiT-B, 12 images (2364 token rows: the weights-in-registers GEMM applies): attn.proj -> attn.qact3 -> Block.qact2 and
    mlp.fc2 -> mlp.qact2 -> Block.qact4 are ONE kernel each (ivit_gemm_i8_requant_residual_ex), 8 launches per block; logits
    bitwise equal to the fused engine's and, on the golden images inside the batch, to the reference's"""
    from ivit_amd.quantization_utils import lazy
    model, meta, z = load_model(tag)
    n = meta["n_images"]
    imgs = np.concatenate([synth.make_images(12 - n, 515), synth.make_images(n, meta["image_seed"])])
    imgs = torch.from_numpy(imgs).to(DEV)
    with torch.no_grad():
        ye = model(imgs)                 # fused engine
        model.use_engine = False
        model(imgs)
        lazy.STATS.update(fused=0, materialised=0)
        torch.cuda.set_sync_debug_mode("error")
        try:
            ym = model(imgs)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    assert lazy.STATS == {"fused": 8 * len(model.blocks) + 3, "materialised": 1}, lazy.STATS
    assert torch.equal(ye, ym)
    if "regime" not in meta:
        assert np.array_equal(bits(ym)[12 - n:], z["logits_f32_bits"])
    assert np.array_equal(ym.argmax(dim=1).cpu().numpy()[12 - n:], z["top1"])


def test_int8_carrying_path_for_a_caller_that_drives_the_modules_itself():
    """the literal drop-in scenario: somebody else's forward (here: the reference's forward_features / forward bodies, vit_quant.py:285-312,
    written out against the sub-modules) calls the modules one by one.  With lazy.enable_everywhere() the frozen QuantActs carry int8
    there too, the result is a tensor that materialises the reference's float logits when looked at, and nothing is read back"""
    from ivit_amd.quantization_utils import lazy
    model, meta, z = load_model("deit_small")
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)

    def reference_forward(m, x):                       # vit_quant.py:285-312
        B = x.shape[0]
        x, s = m.qact_input(x)
        x, s = m.patch_embed(x, s)
        cls = m.cls_token.expand(B, -1, -1)
        x = torch.cat((cls, x), dim=1)
        xp, sp = m.qact_pos(m.pos_embed)
        x, s = m.qact1(x, s, xp, sp)
        x = m.pos_drop(x)
        for blk in m.blocks:
            x, s = blk(x, s)
        x, s = m.norm(x, s)
        x = x[:, 0]
        x, s = m.qact2(x, s)
        x = m.pre_logits(x)
        x, s = m.head(x, s)
        return x

    lazy.enable_everywhere(True)
    try:
        with torch.no_grad():
            reference_forward(model, imgs)             # warm-up
            lazy.STATS.update(fused=0, materialised=0)
            torch.cuda.set_sync_debug_mode("error")
            try:
                y = reference_forward(model, imgs)
            finally:
                torch.cuda.set_sync_debug_mode("default")
        assert isinstance(y, lazy.QT) and lazy.STATS["materialised"] == 0 and lazy.STATS["fused"] == 10 * len(model.blocks) + 3
        top1 = y.argmax(dim=1)                         # looking at it: the float logits
        assert np.array_equal(top1.cpu().numpy(), z["top1"])
        assert np.array_equal(bits(y + 0.0), z["logits_f32_bits"])
    finally:
        lazy.enable_everywhere(False)
    assert not lazy.active()


@pytest.mark.parametrize("img,patch", [(160, 16), (224, 32), (96, 8)])
def test_engine_other_geometries_equal_the_module_path(img, patch):
    """VisionTransformer(img_size, patch_size) of the reference is parametric (vit_quant.py:158-197): the fused engine takes square
    geometries with 3 * patch^2 % 64 == 0 and at most 207 tokens (101, 50 and 145 tokens here), calibrated on the GPU, and equals
    the literal module-by-module path bit for bit (float round trip per module, IVIT_LAZY off) as well as the int8-carrying one"""
    from ivit_amd.quantization_utils import lazy
    torch.manual_seed(img + patch)
    model = ivit.VisionTransformer(img_size=img, patch_size=patch, embed_dim=192, depth=3, num_heads=3, mlp_ratio=4, qkv_bias=True,
                                   num_classes=40).to(DEV).eval()
    with torch.no_grad():
        for p in model.parameters():          # wider weights than the init's 0.02: activations that use their ranges
            if p.dim() > 1:
                p.mul_(3.0)
        g = torch.Generator(device="cpu").manual_seed(5)
        calib = torch.randn(6, 3, img, img, generator=g).to(DEV)
        model(calib)
        model(calib.flip(0) * 0.7)
    ivit.freeze_model(model)
    assert model.engine_unsupported_reason() is None, model.engine_unsupported_reason()
    x = torch.randn(5, 3, img, img, generator=g).to(DEV)
    with torch.no_grad():
        ye = model(x)
        assert model._engine is not None and model._engine[2].T == (img // patch) ** 2 + 1
        model.use_engine = False
        yl = model(x)
        old = lazy.ENABLED
        try:
            lazy.ENABLED = False
            ym = model(x)
        finally:
            lazy.ENABLED = old
    assert torch.equal(ye, ym) and torch.equal(yl, ym)
    assert len(torch.unique(ye.argmax(dim=1))) > 1
    bad = ivit.VisionTransformer(img_size=224, patch_size=14, embed_dim=192, depth=1, num_heads=3)
    assert "geometry" in bad.engine_unsupported_reason()


def test_int8_module_path_with_the_reference_attention_forward(monkeypatch):
    """the reference's Attention.forward takes q, k, v by INDEXING the permuted view (vit_quant.py:65-72) where the mirror unbinds it: the recorded views replay to the same strides, so the head-major GEMM epilogue and the fused attention
    still apply (8 launches per block at >= 2048 token rows) and the logits stay bitwise"""
    from ivit_amd.quantization_utils import lazy
    import ivit_amd.vit_quant as vq

    def attention_forward(self, tokens, s_tokens):
        """the same module calls as the mirror's Attention.forward, with q, k, v taken by integer indexing of the permuted view"""
        batch, n_tok, width = tokens.shape
        heads = self.num_heads
        packed, s_packed = self.qact1(*self.qkv(tokens, s_tokens))
        by_role = packed.reshape(batch, n_tok, 3, heads, width // heads).permute(2, 0, 3, 1, 4)
        query, key, value = by_role[0], by_role[1], by_role[2]                 # indexing, not unbind
        scores, s_scores = self.matmul_1(query, s_packed, key.transpose(-2, -1), s_packed)
        scores, s_scores = self.qact_attn1(scores * self.scale, s_scores * self.scale)
        probs, s_probs = self.int_softmax(scores, s_scores)
        ctx, s_ctx = self.matmul_2(self.attn_drop(probs), s_probs, value, s_packed)
        ctx, s_ctx = self.qact2(ctx.transpose(1, 2).reshape(batch, n_tok, width), s_ctx)
        out, s_out = self.qact3(*self.proj(ctx, s_ctx))
        return self.proj_drop(out), s_out

    model, meta, z = load_model("deit_base")
    n = meta["n_images"]
    imgs = torch.from_numpy(np.concatenate([synth.make_images(12 - n, 77), synth.make_images(n, meta["image_seed"])])).to(DEV)
    model.use_engine = False
    monkeypatch.setattr(vq.Attention, "forward", attention_forward)
    with torch.no_grad():
        model(imgs)
        lazy.STATS.update(fused=0, materialised=0)
        torch.cuda.set_sync_debug_mode("error")
        try:
            y = model(imgs)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    assert lazy.STATS == {"fused": 8 * len(model.blocks) + 3, "materialised": 1}, lazy.STATS
    assert np.array_equal(bits(y)[12 - n:], z["logits_f32_bits"])


def test_int8_module_path_replays_from_a_hip_graph():
    """no read-back means the module-by-module forward of a frozen model can be captured once and replayed (torch.cuda.CUDAGraph =
    a HIP graph): same logits, new input through the static buffer"""
    model, meta, z = load_model("deit_small")
    model.use_engine = False
    n = meta["n_images"]
    static_in = torch.from_numpy(synth.make_images(n, 4321)).to(DEV)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(static_in)              # warm-up: constants cached, allocator primed
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = model(static_in)
        static_in.copy_(torch.from_numpy(synth.make_images(n, meta["image_seed"])).to(DEV))
        graph.replay()
        torch.cuda.synchronize()
    assert np.array_equal(bits(static_out), z["logits_f32_bits"])
    assert np.array_equal(static_out.argmax(dim=1).cpu().numpy(), z["top1"])
