"""CPU tests of the product's host-side logic (i-vit_amd/prepare.py, synth.py, the C-ABI library's
exports) against the oracle and the reference-generated fixtures.  No GPU compute."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest

import ivit_amd
from ivit_amd import _lib, prepare, synth
from ivit_amd.checkpoint import load_synthetic_model
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_kat.npz"))


def test_dyadic_matches_reference_batch_frexp(kat):
    for ci in kat["rq_cases"]:
        c = f"rq{ci}_"
        m, e = prepare.dyadic(kat[c + "pre"], kat[c + "zsf"])
        assert np.array_equal(m.astype(np.float64), kat[c + "m"])
        assert np.array_equal(e, kat[c + "e"])


def test_batch_frexp_function_matches_reference(kat):
    """quantization_utils.quant_utils.batch_frexp (the reference's functional name, quant_utils.py:151-175) on the ratio tensor the
    reference forms: mantissas / exponents of the reference's own batch_frexp (ops_kat.npz)"""
    import torch
    from ivit_amd.quantization_utils import quant_utils as qu
    for ci in kat["rq_cases"]:
        c = f"rq{ci}_"
        ratio = torch.from_numpy(np.atleast_1d(kat[c + "pre"])).double() / torch.tensor([float(kat[c + "zsf"])]).float().double()
        m, e = qu.batch_frexp(ratio.view(1, 1, -1))
        assert m.shape == (1, 1, ratio.numel()) and m.dtype == torch.int64
        assert np.array_equal(m.numpy().reshape(-1).astype(np.float64), kat[c + "m"])
        assert np.array_equal(e.numpy().reshape(-1).astype(np.int32), kat[c + "e"])
    s = qu.symmetric_linear_quantization_params(8, torch.tensor([-0.3]), torch.tensor([1.27]))
    assert s.dtype == torch.float32 and float(s) == float(np.float32(1.27) / np.float32(127.0))
    assert float(qu.symmetric_linear_quantization_params(8, torch.tensor([0.0]), torch.tensor([0.0]))) == float(np.finfo(np.float32).eps)


def test_linear_params_match_reference_quantlinear(kat):
    lp = prepare.LinearParams(kat["lin_W"], kat["lin_b"], kat["lin_sin"])
    assert np.array_equal(lp.sw, kat["lin_sw"])
    assert np.array_equal(lp.W8.astype(np.int32), kat["lin_wint"])
    assert np.array_equal(lp.b32, kat["lin_bint"])
    assert np.array_equal(lp.s_acc, kat["lin_sacc"])


def test_layernorm_params_match_reference(kat):
    for ci in kat["ln_cases"]:
        c = f"ln{ci}_"
        lp = prepare.LayerNormParams(kat[c + "gamma"], kat[c + "beta"], kat[c + "q_sf"])
        assert np.array_equal(lp.s_ln, kat[c + "sln"])
        assert np.array_equal(lp.bias_int, kat[c + "bias_int"])
        mo, eo = orc.dyadic(kat[c + "sln"], kat[c + "q_sf"])
        assert np.array_equal(lp.m.astype(np.float64), mo) and np.array_equal(lp.e, eo)


def test_quant_sym_matches_reference(kat):
    assert np.array_equal(prepare.quant_sym(kat["qs_x"], kat["qs_s"], 8), kat["qs_out"])
    assert prepare.sym_scale(-3.0, 2.0) == orc.sym_scale(-3.0, 2.0)
    assert prepare.sym_scale(0.0, 0.0) == np.finfo(np.float32).eps


@pytest.mark.parametrize("tag", ["deit_tiny", "deit_small"])
def test_synthetic_weights_reproduce_fixture_digests(tag):
    """the synthetic generator + the product's weight quantisation give the integer weights the
    reference derived when the fixtures were made (SHA-256 of weight_integer / bias_integer)"""
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    sha = dict(zip([str(n) for n in z["wint_names"]], [str(d) for d in z["wint_sha"]]))

    def scale(name):
        lo, hi = ranges[name]
        return prepare.sym_scale(lo, hi)

    checks = {"patch_embed.proj": "qact_input", "blocks.0.attn.qkv": "blocks.0.qact1",
              "blocks.3.mlp.fc1": "blocks.3.qact3", "blocks.11.mlp.fc2": "blocks.11.mlp.qact1", "head": "qact2"}
    for lin, qa in checks.items():
        lp = prepare.LinearParams(fs[lin + ".weight"], fs[lin + ".bias"], scale(qa))
        w = lp.W8.astype(np.int32).reshape(fs[lin + ".weight"].shape)
        assert hashlib.sha256(w.tobytes()).hexdigest()[:16] == sha[lin + ".weight_integer"], lin
        assert hashlib.sha256(lp.b32.tobytes()).hexdigest()[:16] == sha[lin + ".bias_integer"], lin


def test_qact_names_cover_fixture_ranges():
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny")
    assert list(ranges) == synth.qact_names(cfg["depth"])
    assert len(ranges) == 137
    for lo, hi in ranges.values():  # pow2-calibrated regime
        p = np.log2(float(hi) / 127.0)
        assert lo == -hi and p == round(p)


def test_images_are_order_independent():
    a = synth.make_images(3, 5)
    b = synth.make_images(1, 5, start=2)
    assert np.array_equal(a[2], b[0])


def test_c_abi_library_exports_every_declared_symbol():
    """include/*.h <-> libivit_hip.so <-> the ctypes table agree (loads the library, calls nothing
    that needs a GPU)."""
    pat = r"\b(?:int|const char\*)\s+(ivit_[a-z0-9_]+)\s*\("
    declared = set(re.findall(pat, open(os.path.join(ROOT, "include", "ivit_hip.h")).read()))
    hooks = set(re.findall(pat, open(os.path.join(ROOT, "include", "ivit_hip_debug.h")).read()))
    assert declared == set(_lib.SIGNATURES) | {"ivit_version", "ivit_last_error_string"}
    assert hooks == set(_lib.LAB_SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.ivit_version() >= 100
    # the product library is stateless as its header says: none of the measurement hooks, no lab kernels
    assert not any(hasattr(L, name) for name in hooks)
    LAB = _lib.lab()
    assert all(hasattr(LAB, name) for name in declared | hooks)
    # argument validation happens before any HIP call: exercise the error path on CPU
    with pytest.raises(_lib.IvitError, match="NULL"):
        _lib.call("ivit_gemm_i8_i32", None, 64, None, 64, None, None, 64, 64, 64, 64, None)
    assert b"NULL" in L.ivit_last_error_string()


def test_engine_refuses_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IvitError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_module_tree_matches_reference_state_dict_schema(golden_dir):
    """keys AND shapes of the nn.Module mirror equal the reference's checkpoints (fixture from the reference)"""
    import json
    sch = json.load(open(os.path.join(golden_dir, "state_dict_schema.json")))
    for name, ref in sch.items():
        factory, _, family = name.partition("@")     # "<factory>@ibert": the same model with the I-BERT operators
        kw = dict(gelu_type=family, softmax_type=family, layernorm_type=family) if family else {}
        mine = {k: list(v.shape) for k, v in getattr(ivit_amd, factory)(**kw).state_dict().items()}
        assert list(mine) == list(ref), name
        assert mine == ref, name
    with pytest.raises(RuntimeError, match="no network"):
        ivit_amd.deit_tiny_patch16_224(pretrained=True)


def test_checkpoint_harness_cpu(tmp_path):
    """load_model / save_checkpoint with the reference's checkpoint layouts (scripts/inference.py:94-224): dict with
    'model' + 'model_config', bare state_dict, scalar buffers of older checkpoints, overrides; no GPU involved."""
    import torch
    import ivit_amd.quantization_utils as q
    from ivit_amd import inference
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny")
    model = ivit_amd.deit_tiny_patch16_224()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct):
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    path = tmp_path / "checkpoint.pth.tar"
    inference.save_checkpoint(model, path, {"model_name": "deit_tiny", "num_classes": 1000, "gelu_type": "ivit",
                                            "softmax_type": "ivit", "layernorm_type": "ivit", "att_block_out_bw": 8},
                              epoch=3, best_acc1=12.5)
    m2 = inference.load_model(path, device="cpu", strict_load=True)
    assert m2.is_frozen() and not m2.training
    assert float(m2.blocks[3].qact2.x_max) == float(ranges["blocks.3.qact2"][1])
    assert torch.equal(m2.head.weight, model.head.weight)
    # bare state_dict + a scalar buffer, non-strict, no warm-up
    sd = model.state_dict()
    sd["qact_input.x_min"] = sd["qact_input.x_min"].reshape(())
    bare = tmp_path / "bare.pth"
    torch.save(sd, bare)
    with pytest.raises(KeyError, match="ppoly"):          # the reference's default for a config-less checkpoint (:160-162)
        inference.load_model(bare, device="cpu", strict_load=False)
    m3 = inference.load_model(bare, device="cpu", strict_load=False, gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit")
    assert m3.qact_input.x_min.shape == (1,) and m3.is_frozen()
    # model_name selects the factory; unknown operator families are refused
    assert type(inference.build_model({"model_name": "swin_tiny_patch4_window7_224", "gelu_type": "ivit", "softmax_type": "ivit",
                                       "layernorm_type": "ivit"})).__name__ == "SwinTransformer"
    small = inference.build_model({"model_name": "deit_small"})
    assert small.embed_dim == 384 and small.op_types == ("ibert", "ibert", "ibert")     # the reference's defaults (:111-113)
    assert type(inference.build_model({"model_name": "deit_tiny", "gelu_type": "ibert"}).blocks[0].mlp.act).__name__ == "IBERTIntGELU"
    with pytest.raises(KeyError):
        inference.build_model({"model_name": "deit_tiny", "gelu_type": "ppoly_deg_2_seg_16"})
    # evaluate_dataset arithmetic on a stub model
    class Stub(torch.nn.Module):
        def forward(self, x):
            return x
    logits = torch.eye(10)[:6] * 5 + torch.arange(10) * 0.01
    t1, t3, t5 = inference.evaluate_dataset(Stub(), [(logits, torch.tensor([0, 1, 2, 9, 8, 0]))], "cpu",
                                            print_batch_stats=False)
    assert (round(t1, 3), round(t3, 3), round(t5, 3)) == (50.0, 83.333, 83.333)


def test_integer_export_matches_reference_integers(tmp_path):
    """Row f2: the exported int8 weights / int32 biases (TVM_benchmark/convert_model.py:12-66 key names) carry exactly
    the integers the reference derived (SHA-256 in the fixture), survive the params.npy round trip, and the
    ExportSource rebuilt from the two files yields the same engine constants as the float checkpoint."""
    from ivit_amd import export
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_tiny")
    depth = cfg["depth"]
    params, qconfig = export.export_integer_params(fs, ranges, depth)
    sha = dict(zip([str(n) for n in z["wint_names"]], [str(d) for d in z["wint_sha"]]))
    ren = {"patch_embed.proj": "embed_conv_", "head": "head_"}
    for i in range(depth):
        for lin in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            ren[f"blocks.{i}.{lin}"] = f"block_{i}_{lin.replace('.', '_')}_"
    checked = 0
    for ref_name, new in ren.items():
        w, b = params[new + "weight"], params[new + "bias"]
        assert w.dtype == np.int8 and b.dtype == np.int32
        assert hashlib.sha256(w.astype(np.int32).tobytes()).hexdigest()[:16] == sha[ref_name + ".weight_integer"]
        assert hashlib.sha256(b.reshape(-1).tobytes()).hexdigest()[:16] == sha[ref_name + ".bias_integer"]
        checked += 2
    assert checked == len(sha)
    assert params["embed_conv_weight"].shape == (192, 3, 16, 16) and params["embed_conv_bias"].shape == (1, 192, 1, 1)
    assert params["block_0_norm1_bias"].dtype == np.int32 and params["pos_embed_weight"].shape == (1, 197, 192)
    assert set(qconfig) >= {"qconfig_pos", "qconfig_addpos", "qconfig_embed_conv", "block_11_qconfig_add2", "qconfig_norm",
                            "qconfig_head"} and len([k for k in qconfig if "qconfig" in k]) == 5 + 12 * depth
    e = qconfig["block_2_qconfig_qkv"]
    assert np.array_equal(e["output_scale"], (np.float32(e["input_scale"]) * e["kernel_scale"]).astype(np.float32))
    assert qconfig["block_0_qconfig_add1"]["input_dtype"] == "int16" and e["from_scale"] == 65.0

    export.save_export(params, qconfig, str(tmp_path))
    p2, q2 = export.load_export(str(tmp_path))
    assert sorted(p2) == sorted(params) and all(np.array_equal(p2[k], params[k]) for k in params)
    a, b = export.FloatSource(fs, ranges), export.ExportSource(p2, q2)
    for name, qa in (("patch_embed.proj", "qact_input"), ("blocks.5.mlp.fc2", "blocks.5.mlp.qact1"), ("head", "qact2")):
        s_in = a.act_scale(qa)
        assert s_in == b.act_scale(qa)
        la, lb = a.linear(name, s_in), b.linear(name, s_in)
        assert np.array_equal(la.W8, lb.W8) and np.array_equal(la.b32, lb.b32) and np.array_equal(la.s_acc, lb.s_acc)
        s_out = np.float32(2.0 ** -3)
        assert all(np.array_equal(x, y) for x, y in zip(la.requant_to(s_out), lb.requant_to(s_out)))
    na, nb = a.layernorm("blocks.7.norm2", np.float32(0.03125)), b.layernorm("blocks.7.norm2", np.float32(0.03125))
    assert np.array_equal(na.bias_int, nb.bias_int) and np.array_equal(na.s_ln, nb.s_ln) and np.array_equal(na.m, nb.m)
    assert np.array_equal(a.tensor("cls_token"), b.tensor("cls_token"))


def test_operator_signatures_match_reference_dropin_fixture(golden_dir):
    """tests/golden/dropin_signatures.json is written by oracle/dropin_proof.py, which runs the REFERENCE's own
    models/vit_quant.py and swin_quant.py on top of this package's modules in the build container (construction,
    state_dict == the reference's, load_state_dict, freeze / unfreeze).  Here, without the reference: every constructor /
    forward parameter the reference's operator classes have exists in the build's class under the same name, in the same
    order, with the same default (the build may only add trailing parameters)."""
    import inspect
    import ivit_amd.layers_quant as lq
    import ivit_amd.model_utils as mu
    import ivit_amd.quantization_utils as qu
    fx = json.load(open(os.path.join(golden_dir, "dropin_signatures.json")))
    assert set(fx["reference_files_on_build_modules"]) >= {"deit_tiny_patch16_224", "deit_base_patch16_224",
                                                           "swin_tiny_patch4_window7_224", "deit_tiny_patch16_224@ibert"}

    def sig(obj):
        return [[n, None if p.default is inspect.Parameter.empty else repr(p.default)]
                for n, p in inspect.signature(obj).parameters.items() if n not in ("self", "args", "kwargs")]

    def compatible(mine, ref, what):
        assert len(mine) >= len(ref), what
        for (n1, d1), (n0, d0) in zip(mine, ref):
            assert n1 == n0, (what, n1, n0)
            assert d1 == d0 or (d0 is not None and d1 is not None and d0.split(".")[-1] == d1.split(".")[-1]), (what, n1, d1, d0)

    for name, spec in fx["signatures"].items():
        obj = getattr(qu, name, None) or getattr(qu.quant_utils, name, None) or getattr(lq, name, None) or getattr(mu, name, None)
        assert obj is not None, name
        if "call" in spec:
            compatible(sig(obj), spec["call"], name)
            continue
        if "autograd_forward" in spec:     # torch.autograd.Function of quant_utils.py: forward(ctx, ...)
            compatible(sig(obj.forward), spec["autograd_forward"], name + ".forward")
            assert callable(obj.apply)
            continue
        compatible(sig(obj.__init__), spec["init"], name + ".__init__")
        compatible(sig(obj.forward), spec["forward"], name + ".forward")
        for m in spec["methods"]:
            assert callable(getattr(obj, m, None)), (name, m)


def test_int8_carrying_tensor_mechanics():
    """quantization_utils/lazy.py on the CPU (no kernel runs): a QT answers shape questions like the float tensor it stands for,
    the model files' shape operations act on its int8 payload or are recorded on its pending node, a QS carries its host value
    through `* python scalar` in float32, and anything else materialises `integer * scale`."""
    import torch
    import torch.nn.functional as F
    from ivit_amd.quantization_utils import lazy
    QT, QS = lazy.QT, lazy.QS
    rng = np.random.default_rng(3)
    q = torch.from_numpy(rng.integers(-128, 128, (2, 5, 24), dtype=np.int8))
    s = QS.make(0.03, "cpu")
    x = QT.wrap(q.shape, q.device, q8=q, scale=s)
    assert x.shape == (2, 5, 24) and x.dim() == 3 and x.dtype == torch.float32 and not x.is_cuda and x.numel() == 240
    a, b, c = x.reshape(2, 5, 3, 2, 4).permute(2, 0, 3, 1, 4).unbind(0)              # vit_quant.py:66-68
    assert isinstance(a, QT) and a.shape == (2, 2, 5, 4) and torch.equal(b.q8, q.reshape(2, 5, 3, 2, 4).permute(2, 0, 3, 1, 4)[1])
    assert b.transpose(-2, -1).shape == (2, 2, 4, 5)
    f = x.to_float()
    assert f.dtype == torch.float32 and torch.equal(f, q.float() * np.float32(0.03))
    assert F.dropout(x, 0.1, False) is x
    y = x * 0.125
    assert isinstance(y, QT) and y.q8 is x.q8 and y.scale.host[0] == np.float32(0.03) * np.float32(0.125)
    assert type(x + 1) is torch.Tensor and torch.equal(x + 1, f + 1)                  # not part of the protocol: floats
    s2 = s * 0.125
    assert isinstance(s2, QS) and (s * 0.125) is s2 and s2.host[0] == np.float32(0.03) * np.float32(0.125)
    assert float(s2) == float(np.float32(0.03) * np.float32(0.125))
    assert s.view(1, -1).host is s.host and type(s + s) is torch.Tensor
    # a pending node: the views are recorded and replayed on the materialised tensor
    n = QT.wrap(x.shape, "cpu", node=lazy.Scaled(x, 2.0))
    v = n[:, 0]
    assert v.shape == (2, 24) and torch.equal(v.to_float(), (f * 2.0)[:, 0])
    t1, t2 = n.transpose(1, 2).unbind(0)
    assert t2.shape == (24, 5) and torch.equal(t2.to_float(), (f * 2.0).transpose(1, 2)[1])
    with lazy.scope(True):
        cat = torch.cat((torch.ones(2, 1, 24), x), dim=1)                             # vit_quant.py:293
    assert isinstance(cat, QT) and cat.shape == (2, 6, 24) and torch.equal(cat.to_float(), torch.cat((torch.ones(2, 1, 24), f), 1))
    assert not lazy.active()


def test_lazy_scope_depth_is_per_thread_and_survives_enable_everywhere():
    """quantization_utils/lazy.py: `enable_everywhere(False)` inside an open scope must not corrupt the scope depth (round-3
    advisor finding: it reset the counter, the scope's exit then drove it to -1 and the int8-carrying path stayed off for the
    rest of the process), and a scope opened on one thread is not seen by another."""
    import threading
    from ivit_amd.quantization_utils import lazy
    if not lazy.ENABLED:
        pytest.skip("IVIT_LAZY=0")
    assert not lazy.active()
    with lazy.scope(True):
        assert lazy.active()
        lazy.enable_everywhere(True)
        lazy.enable_everywhere(False)
        assert lazy.active()                      # the scope is still open
        seen = []
        t = threading.Thread(target=lambda: seen.append(lazy.active()))
        t.start(); t.join()
        assert seen == [False]                    # another thread: no scope of its own
    assert not lazy.active()
    with lazy.scope(True):                        # and the next scope works as the first did
        assert lazy.active()
    assert not lazy.active()
    lazy.enable_everywhere(True)
    try:
        assert lazy.active()
    finally:
        lazy.enable_everywhere(False)
    assert not lazy.active()


def test_bench_numa_cpus_of_gpu_on_a_fake_sysfs_tree(tmp_path, monkeypatch):
    """bench.numa_cpus_of_gpu (the ranks of `bench.py --gpus N` pin themselves to their GPU's NUMA node): PCI order = HIP order,
    HIP_VISIBLE_DEVICES remaps, numa_node = -1 / a node outside the affinity mask / a missing tree say WHY there is no answer"""
    import bench
    root = tmp_path / "sys"

    def gpu(card, bdf, node, vendor="0x1002", cls="0x120000"):
        dev = root / "devices" / "pci0000:00" / bdf
        dev.mkdir(parents=True)
        (dev / "vendor").write_text(vendor + "\n"); (dev / "class").write_text(cls + "\n"); (dev / "numa_node").write_text(f"{node}\n")
        d = root / "class" / "drm" / card
        d.mkdir(parents=True)
        (d / "device").symlink_to(dev)

    gpu("card1", "0000:85:00.0", 1)           # enumerated second by HIP (higher bus address)
    gpu("card0", "0000:05:00.0", 0)
    gpu("card2", "0000:03:00.0", 0, vendor="0x1a03", cls="0x030000")      # the BMC's VGA: not an AMD GPU
    gpu("card3", "0000:c5:00.0", -1)
    xcp = root / "devices" / "platform" / "amdgpu_xcp_22"          # a partition node: a drm card without PCI attributes
    xcp.mkdir(parents=True)
    (root / "class" / "drm" / "card9").mkdir(parents=True)
    (root / "class" / "drm" / "card9" / "device").symlink_to(xcp)
    for node, cpus in ((0, "0-3,16-19"), (1, "4-7")):
        n = root / "devices" / "system" / "node" / f"node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpus + "\n")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False); monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    every = set(range(32))
    assert bench.numa_cpus_of_gpu(0, str(root), every) == {0, 1, 2, 3, 16, 17, 18, 19}
    assert bench.numa_cpus_of_gpu(1, str(root), every) == {4, 5, 6, 7}
    assert bench.numa_cpus_of_gpu(0, str(root), {2, 3, 4}) == {2, 3}            # clipped to the affinity mask
    why = []
    assert bench.numa_cpus_of_gpu(2, str(root), every, why) is None and "numa_node = -1" in why[0]
    why = []
    assert bench.numa_cpus_of_gpu(5, str(root), every, why) is None and "3 GPU(s)" in why[0]
    why = []
    assert bench.numa_cpus_of_gpu(1, str(root), {0, 1}, why) is None and "affinity mask" in why[0]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert bench.numa_cpus_of_gpu(0, str(root), every) == {4, 5, 6, 7}
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    why = []
    assert bench.numa_cpus_of_gpu(0, str(tmp_path / "nothing"), every, why) is None and "no AMD GPU" in why[0]


# ----------------------------------------------------------------------------------- csrc/check_isa.py (build-time ISA lint)
def _isa(body):
    """a guarded kernel's assembly text around `body` (lines; `A:` marks an inline-asm statement)"""
    out = ["\t.text", "_ZN12_GLOBAL__N_119gemm_i8_wreg_kernelILi0ELi0ELb1EEEvNS_8GemmArgsE:"]
    for ln in body:
        if ln.startswith("A:"):
            out += ["\t;;#ASMSTART", "\t" + ln[2:].strip(), "\t;;#ASMEND"]
        elif ln.endswith(":"):
            out.append(ln)
        else:
            out.append("\t" + ln)
    out += ["\ts_endpgm", "\t.section\t.rodata"]
    return "\n".join(out) + "\n"


def _lint(body):
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "i-vit_amd", "csrc", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.check_text(_isa(body))[0]


def test_isa_lint_accepts_counted_waits_and_rejects_broken_streams():
    """csrc/check_isa.py on hand-made instruction streams: the disciplines of the inline-asm GEMM kernels hold in the accepted
    ones and each rejected one breaks exactly one of them (the build runs the same checker on the real gemm.s)"""
    load_a, load_b = "A: global_load_dwordx4 v[10:13], v[2:3], off", "A: global_load_dwordx4 v[14:17], v[2:3], off offset:1024"
    mfma = "A: v_mfma_i32_16x16x64_i8 v[40:43], v[10:13], v[20:23], v[40:43]"
    # (i) a use behind a covering wait; the younger load may stay in flight
    assert _lint([load_a, load_b, "A: s_waitcnt vmcnt(1)", "v_add_u32_e32 v30, v10, v11"]) == []
    # ... the same use with the count one too lenient
    bad = _lint([load_a, load_b, "A: s_waitcnt vmcnt(2)", "v_add_u32_e32 v30, v10, v11"])
    assert len(bad) == 1 and "rule i" in bad[0] and "v10" in bad[0]
    # ... a compiler-scheduled copy between the load and its wait (the hazard class of round 3), reads and writes alike
    assert any("rule i" in f for f in _lint([load_a, "v_mov_b32_e32 v50, v12", "A: s_waitcnt vmcnt(0)"]))
    assert any("rule i" in f for f in _lint([load_a, "v_mov_b32_e32 v12, v50", "A: s_waitcnt vmcnt(0)"]))
    # ... every vector-memory operation counts for vmcnt, whoever issued it: a store behind the loads is one more younger operation
    assert _lint([load_a, "global_store_dword v[4:5], v60, off", load_b, "A: s_waitcnt vmcnt(2)", "v_mov_b32_e32 v50, v10"]) == []
    assert _lint([load_a, load_b, "global_store_dword v[4:5], v60, off", "A: s_waitcnt vmcnt(1)", "v_mov_b32_e32 v50, v16"]) == []
    assert any("rule i" in f for f in _lint([load_a, load_b, "global_store_dword v[4:5], v60, off", "A: s_waitcnt vmcnt(2)",
                                             "v_mov_b32_e32 v50, v16"]))
    # ... loads in flight around a loop's back edge: issued in one iteration, waited for at the top of the next
    loop_ok = [load_a, ".LBB0_1:", "A: s_waitcnt vmcnt(0)", "v_mov_b32_e32 v50, v10", load_a, "s_cbranch_scc1 .LBB0_1", "A: s_waitcnt vmcnt(0)"]
    assert _lint(loop_ok) == []
    loop_bad = [load_a, ".LBB0_1:", "v_mov_b32_e32 v50, v10", "A: s_waitcnt vmcnt(0)", load_a, "s_cbranch_scc1 .LBB0_1", "A: s_waitcnt vmcnt(0)"]
    assert any("rule i" in f for f in _lint(loop_bad))
    # ... ds_read through lgkmcnt, and a scalar load sneaking into its window
    ds = "A: ds_read_b128 v[20:23], v5 offset:1024"
    assert _lint([ds, "A: s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v60, v21"]) == []
    assert any("rule i" in f for f in _lint([ds, "v_mov_b32_e32 v60, v21", "A: s_waitcnt lgkmcnt(0)"]))
    assert any("rule i-b" in f for f in _lint([ds, "s_load_dwordx2 s[4:5], s[0:1], 0x0", "A: s_waitcnt lgkmcnt(0)"]))
    # (ii) an inline-asm MFMA's accumulator read by a VALU instruction too early / after the kernel's two s_nop 15
    pre = [load_a, "A: s_waitcnt vmcnt(0)", ds, "A: s_waitcnt lgkmcnt(0)"]
    assert any("rule ii" in f for f in _lint(pre + [mfma, "s_nop 3", "v_cvt_f32_i32_e32 v60, v41"]))
    assert _lint(pre + [mfma, "s_nop 15", "s_nop 15", "v_cvt_f32_i32_e32 v60, v41"]) == []
    assert any("rule ii" in f for f in _lint(pre + [mfma, mfma]))          # back to back on one accumulator
    # (iii) an inline-asm store without its s_nop
    assert any("rule iii" in f for f in _lint(["A: global_store_dwordx4 v[4:5], v[60:63], off"]))
    # (iv) a VALU write of an MFMA operand right in front of the inline-asm MFMA (round 4: the bias copy sunk to the tile's first MFMA
    #      left half of the accumulator stale); far enough in front, or another register, is fine
    assert any("rule iv" in f for f in _lint(pre + ["v_mov_b64_e32 v[42:43], v[68:69]", "v_mov_b64_e32 v[40:41], v[66:67]", mfma]))
    assert _lint(pre + ["v_mov_b64_e32 v[42:43], v[68:69]", "v_mov_b64_e32 v[40:41], v[66:67]", "s_nop 0", "v_add_u32_e32 v70, v71, v72",
                        "v_add_u32_e32 v73, v71, v72", "v_add_u32_e32 v74, v71, v72", mfma]) == []
    assert _lint(pre + ["v_mov_b64_e32 v[44:45], v[68:69]", mfma]) == []


def test_window_shiftexp_band_conditions():
    """prepare.window_shiftexp_band (round 4): when may the Swin window attention read Shiftmax's exp_int from a table?  One row when
    it depends on the distance to the maximum alone, 256 rows otherwise, nothing when a score under the shift mask could be a row
    maximum or fails to saturate (-100 / s too small) or the band is wider than the kernel's LDS slices"""
    import numpy as np
    from ivit_amd.prepare import phi_table, shiftexp2d, window_shiftexp_band
    band, W = window_shiftexp_band(0.271, True)
    assert band is not None and band.shape == (1, W) and W % 16 == 0 and W <= 192
    tab = shiftexp2d(np.float32(0.271))
    for qmax in (0, 100, 255):                      # the one row reproduces every row of the full table, saturated beyond W - 1
        for j in range(0, min(qmax + 1, 256)):
            assert band[0, min(j, W - 1)] == tab[qmax, qmax - j]
    band, W = window_shiftexp_band(0.1173, True)
    assert band is not None and band.shape == (256, W)
    assert np.all(band[:, W - 1] == band[0, W - 1])
    assert window_shiftexp_band(1.3, True) == (None, 0)             # -100 / 1.3 = -77: a masked score can exceed an unmasked one
    band, W = window_shiftexp_band(1.3, False)
    assert band is not None                                         # ... without a mask the table is fine
    assert window_shiftexp_band(0.01, False) == (None, 0)           # band wider than 192 entries (or than the table)
    assert np.array_equal(phi_table(0.25), np.arange(-128, 128, dtype=np.float32))



def test_serial_load_scan_flags_a_wait_between_independent_loads(tmp_path):
    """scripts/scan_serial_loads.py (advisory ISA scan): a kernel whose loads are separated by s_waitcnt vmcnt(0) is
    listed, one whose loads are issued together and waited for once is not"""
    import subprocess
    import sys
    serial = "\n".join(["global_load_dword v1, v[2:3], off", "s_waitcnt vmcnt(0)", "ds_write_b32 v9, v1"] * 3)
    batched = "\n".join(["global_load_dword v%d, v[2:3], off" % i for i in range(1, 4)] + ["s_waitcnt vmcnt(0)", "ds_write_b32 v9, v1"])
    text = ""
    for name, body in (("_Z6serialv", serial), ("_Z7batchedv", batched)):
        text += f"\t.type\t{name},@function\n{name}:\n{body}\n\ts_endpgm\n.Lfunc_end_{name}:\n"
    src = tmp_path / "k.s"
    src.write_text(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "scan_serial_loads.py"), str(src)], capture_output=True, text=True, check=True).stdout
    assert "_Z6serialv" in out and "_Z7batchedv" not in out
