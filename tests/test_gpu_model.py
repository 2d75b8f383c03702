"""GPU parity, whole model: the HIP engine against the golden vectors produced by the REFERENCE
(tests/golden/*.npz: INT32 logits, top-1 and CRC32 of intermediate taps) and against the CPU oracle."""
import zlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ivit = pytest.importorskip("ivit_amd")
from ivit_amd import synth  # noqa: E402
from ivit_amd.checkpoint import load_synthetic_model  # noqa: E402
from ivit_amd.engine import IntViTEngine  # noqa: E402

DEV = "cuda:0"


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.int32).tobytes())


def build(tag, max_batch):
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=max_batch)
    return eng, fs, ranges, cfg, meta, z


@pytest.mark.parametrize("tag", ["deit_tiny", "deit_small", "deit_base", "vit_base", "vit_large"])
def test_golden_logits_and_taps(tag):
    eng, fs, ranges, cfg, meta, z = build(tag, meta_batch(tag))
    n = meta["n_images"]
    imgs = torch.from_numpy(synth.make_images(n, meta["image_seed"])).to(DEV)
    taps = {}
    li, lf, t1 = eng.forward(imgs, taps)
    torch.cuda.synchronize()
    # every tap the reference recorded and the engine materialises, by CRC32
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    checked = 0
    for name, t in taps.items():
        if name in gold:
            assert crc(t.cpu().numpy().astype(np.int32)) == int(gold[name]), f"{tag}: tap {name} differs"
            checked += 1
    assert checked == 7 * cfg["depth"] + 3  # 7 QuantAct taps per block + stem + tail
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"]), f"{tag}: INT32 logits differ"
    assert np.array_equal(lf.cpu().numpy().view(np.int32), z["logits_f32_bits"])
    assert np.array_equal(t1.cpu().numpy().astype(np.int64), z["top1"])


@pytest.mark.parametrize("tag", ["deit_tiny_ibert", "deit_tiny_ibert_natural"])
def test_ibert_engine_golden_logits_and_taps(tag):
    """the fork's default operator family (I-BERT GELU / Softmax / LayerNorm) in the fused engine, power-of-two ranges and
    ranges as calibrated: every QuantAct tap, INT32 logits and top-1 of the REFERENCE"""
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=4, family="ibert")
    n = meta["n_images"]
    imgs = torch.from_numpy(synth.make_images(n, meta["image_seed"])).to(DEV)
    taps = {}
    li, lf, t1 = eng.forward(imgs, taps)
    torch.cuda.synchronize()
    gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
    bad = [name for name, t in taps.items() if name in gold and crc(t.cpu().numpy().astype(np.int32)) != int(gold[name])]
    checked = sum(1 for name in taps if name in gold)
    assert not bad, (bad[:5], checked)
    assert checked == 7 * cfg["depth"] + 3
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"]), f"{tag}: INT32 logits differ"
    assert np.array_equal(t1.cpu().numpy().astype(np.int64), z["top1"])
    if "regime" not in meta:
        assert np.array_equal(lf.cpu().numpy().view(np.int32), z["logits_f32_bits"])


def meta_batch(tag):
    return {"deit_tiny": 8, "deit_small": 4, "deit_base": 4, "vit_base": 2, "vit_large": 2}[tag]


def test_full_taps_deit_tiny():
    """the complete stage-by-stage tensors of image 0 stored in the fixture"""
    eng, fs, ranges, cfg, meta, z = build("deit_tiny", 1)
    imgs = torch.from_numpy(synth.make_images(1, meta["image_seed"])).to(DEV)
    taps = {}
    eng.forward(imgs, taps)
    n = 0
    for key in z.files:
        if key.startswith("full/") and key[5:] in taps:
            got = taps[key[5:]].cpu().numpy().astype(np.int32)
            exp = z[key].astype(np.int32)
            assert np.array_equal(got.reshape(exp.shape), exp), key
            n += 1
    assert n >= 10


def test_batch_invariance_and_oracle_on_fresh_images():
    """Config 2 shape (DeiT-S, batch 64): images the fixtures never saw, checked against the oracle on a
    subset, and the size-independent property that an image's logits do not depend on its batch."""
    eng, fs, ranges, cfg, meta, z = build("deit_small", 64)
    imgs_np = synth.make_images(64, 4242)
    imgs = torch.from_numpy(imgs_np).to(DEV)
    li, lf, t1 = eng.forward(imgs)
    li = li.cpu().numpy().copy()
    om = orc.OracleViT(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"])
    sub = [0, 17, 63]
    ref = om.forward(imgs_np[sub])
    assert np.array_equal(li[sub], ref["logits_int32"])
    perm = np.random.default_rng(0).permutation(64)
    li2, _, _ = eng.forward(imgs[torch.from_numpy(perm).to(DEV)].contiguous())
    assert np.array_equal(li2.cpu().numpy(), li[perm])
    li3, _, _ = eng.forward(imgs[5:6].contiguous())
    assert np.array_equal(li3.cpu().numpy(), li[5:6])


@pytest.mark.parametrize("tag,family", [("deit_base", "ivit"), ("deit_base_natural", "ivit")])
def test_logits_do_not_depend_on_the_batch_size(tag, family):
    """DeiT-B through every kernel-selection regime of the engine: batch 1 .. 10 (small-tile GEMMs, row-major operands), 11, 12
    (first persistent / weights-in-registers launches, ragged 128-token tiles: 2167 and 2364 rows), 37, 130 (partial last
    tiles, half-tile tail or not) -- an image's INT32 logits are the same in all of them"""
    fs, ranges, cfg, meta, z = load_synthetic_model(tag)
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=130, family=family)
    imgs = torch.from_numpy(synth.make_images(130, 777)).to(DEV)
    ref = eng.forward(imgs)[0].cpu().numpy().copy()
    assert len(set(ref.argmax(1).tolist())) > 10
    for n in (1, 3, 10, 11, 12, 37):
        got = eng.forward(imgs[:n].contiguous())[0].cpu().numpy()
        assert np.array_equal(got, ref[:n]), n
    got = eng.forward(imgs[93:130].contiguous())[0].cpu().numpy()
    assert np.array_equal(got, ref[93:130])


def test_block_layout_path_equals_row_major_path():
    """batch large enough for the persistent GEMM (M = 13 * 197 >= 2048, not a multiple of 16: the last 16-row block of
    every block-layout operand is partly padding): activations + weights in the block layout vs everything row-major --
    identical logits AND identical stage-by-stage taps (the block-layout taps are untiled for the caller)"""
    eng, fs, ranges, cfg, meta, z = build("deit_tiny", 13)
    imgs = torch.from_numpy(synth.make_images(13, 99)).to(DEV)
    assert eng.block_operands
    t_blocks, t_rows = {}, {}
    li_b = eng.forward(imgs, t_blocks)[0].cpu().numpy().copy()
    eng.block_operands = False
    li_r = eng.forward(imgs, t_rows)[0].cpu().numpy().copy()
    eng.block_operands = True
    assert np.array_equal(li_b, li_r)
    assert set(t_blocks) == set(t_rows) and len(t_rows) > 50
    for k in t_rows:
        assert torch.equal(t_blocks[k], t_rows[k]), k


def test_aliased_workspaces_equal_separate_buffers():
    """the engine's workspace aliasing (in-place residual QuantActs and GELU, attention output over the LayerNorm buffer,
    q/k/v inside the fc1 buffer) changes no bit: logits and taps against every intermediate in its own buffer"""
    eng, fs, ranges, cfg, meta, z = build("deit_small", 16)
    imgs = torch.from_numpy(synth.make_images(16, 4321)).to(DEV)
    t_alias, t_own = {}, {}
    li_a = eng.forward(imgs, t_alias)[0].cpu().numpy().copy()
    eng._compact(False)
    eng.gelu_in_place = False
    li_o = eng.forward(imgs, t_own)[0].cpu().numpy().copy()
    eng._compact(True)
    eng.gelu_in_place = True
    assert np.array_equal(li_a, li_o)
    assert set(t_alias) == set(t_own)
    for k in t_own:
        assert torch.equal(t_alias[k], t_own[k]), k


def test_headline_batch_256_deit_base():
    """Config 3 at full size: golden images embedded in a batch of 256 reproduce the golden logits."""
    eng, fs, ranges, cfg, meta, z = build("deit_base", 256)
    imgs_np = synth.make_images(256, 777)
    gold = synth.make_images(meta["n_images"], meta["image_seed"])
    pos = [0, 100, 200, 255][: meta["n_images"]]
    for p, g in zip(pos, gold):
        imgs_np[p] = g
    li, lf, t1 = eng.forward(torch.from_numpy(imgs_np).to(DEV))
    li = li.cpu().numpy()
    assert np.array_equal(li[pos], z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy()[pos].astype(np.int64), z["top1"])
    # determinism
    li_b, _, _ = eng.forward(torch.from_numpy(imgs_np).to(DEV))
    assert np.array_equal(li_b.cpu().numpy(), li)


def test_config4_shard_batch_128_vit_base():
    """Config 4's per-rank shard at full size (ViT-B, 1024 images over 8 GPUs = 128 per rank): golden images embedded in a
    batch of 128 reproduce the reference's INT32 logits; permutation invariance at that size; fresh images vs the oracle."""
    eng, fs, ranges, cfg, meta, z = build("vit_base", 128)
    imgs_np = synth.make_images(128, 4128)
    gold = synth.make_images(meta["n_images"], meta["image_seed"])
    pos = [3, 126][: meta["n_images"]]
    for p, g in zip(pos, gold):
        imgs_np[p] = g
    imgs = torch.from_numpy(imgs_np).to(DEV)
    li, lf, t1 = eng.forward(imgs)
    li = li.cpu().numpy().copy()
    assert np.array_equal(li[pos], z["logits_int32"])
    assert np.array_equal(t1.cpu().numpy()[pos].astype(np.int64), z["top1"])
    perm = np.random.default_rng(4).permutation(128)
    li2, _, _ = eng.forward(imgs[torch.from_numpy(perm).to(DEV)].contiguous())
    assert np.array_equal(li2.cpu().numpy(), li[perm])
    om = orc.OracleViT(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"])
    assert np.array_equal(li[[64]], om.forward(imgs_np[[64]])["logits_int32"])


def test_engine_from_integer_export(tmp_path):
    """Row f2: params.npy + qconfig.npy (integer weights and scale table only) rebuild an engine whose logits equal the
    reference golden; and the module mirror's state_dict exports the same integers through the reference's route."""
    from ivit_amd import export
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_small")
    params, qconfig = export.export_integer_params(fs, ranges, cfg["depth"])
    export.save_export(params, qconfig, str(tmp_path))
    p2, q2 = export.load_export(str(tmp_path))
    eng = IntViTEngine(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], device=DEV, max_batch=4,
                       source=export.ExportSource(p2, q2))
    imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
    li, lf, t1 = eng.forward(imgs)
    assert np.array_equal(li.cpu().numpy(), z["logits_int32"])
    assert np.array_equal(lf.cpu().numpy().view(np.int32), z["logits_f32_bits"])
    # state_dict route (convert_model.py:12-66 on the module mirror after a frozen module-path forward)
    import ivit_amd as ivit
    import ivit_amd.quantization_utils as q
    model = ivit.deit_small_patch16_224()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    for name, mod in model.named_modules():
        if isinstance(mod, q.QuantAct):
            mod.x_min.fill_(float(ranges[name][0]))
            mod.x_max.fill_(float(ranges[name][1]))
    model.to(DEV)
    ivit.freeze_model(model)
    model.use_engine = False
    with torch.no_grad():
        model(imgs[:1])
    out = export.save_params_from_state_dict(model.state_dict(), cfg["depth"], str(tmp_path / "sd"))
    assert sorted(out) == sorted(params)
    for k in params:
        assert out[k].dtype == params[k].dtype and np.array_equal(out[k], params[k]), k


def test_graph_replay_matches_eager():
    """HIP-graph replay of the forward (launch-bound shapes) returns the same logits as the eager launches, also after
    the inputs change and for a second batch size"""
    eng, fs, ranges, cfg, meta, z = build("deit_tiny", 8)
    imgs = torch.from_numpy(synth.make_images(8, meta["image_seed"])).to(DEV)
    li, _, _ = eng.forward(imgs)
    want = li.cpu().numpy().copy()
    assert np.array_equal(want, z["logits_int32"])
    g1, _, _ = eng.forward_graph(imgs)
    assert np.array_equal(g1.cpu().numpy(), want)
    perm = torch.arange(7, -1, -1, device=DEV)
    g2, _, t2 = eng.forward_graph(imgs[perm].contiguous())
    assert np.array_equal(g2.cpu().numpy(), want[::-1])
    g3, _, _ = eng.forward_graph(imgs[:1].contiguous())
    assert np.array_equal(g3.cpu().numpy(), want[:1])


def test_uint8_input_equals_the_float_pipeline():
    """uint8 pixels through ivit_quantize_patchify_u8_i8 (a 3 x 256 table of what ToTensor + Normalize + the input QuantAct make
    of every (channel, pixel value)) == the float32 images the reference's data pipeline would hand over (v / 255, (x - mean) / std
    in float32, then the float path): the int8 patch operand and the logits, eager and HIP-graph replay"""
    from ivit_amd.prepare import IMAGENET_MEAN, IMAGENET_STD, input_lut_u8
    eng = build("deit_tiny", 5)[0]
    rng = np.random.default_rng(77)
    u8 = torch.from_numpy(rng.integers(0, 256, size=(5, 3, 224, 224), dtype=np.uint8)).to(DEV)
    u8[0, 0, :16, :16] = 0
    u8[0, 1, :16, :16] = 255
    mean = torch.tensor(IMAGENET_MEAN, device=DEV).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=DEV).view(1, 3, 1, 1)
    xf = ((u8.float().div(255) - mean) / std).contiguous()          # torchvision ToTensor + Normalize
    li_f, lf_f, t_f = [t.clone() for t in eng.forward(xf)]
    a0_f = eng.ws["a0"][: 5 * 196].clone()
    li_u, lf_u, t_u = [t.clone() for t in eng.forward(u8)]
    assert torch.equal(eng.ws["a0"][: 5 * 196], a0_f) and a0_f.abs().max() > 40
    assert torch.equal(li_u, li_f) and torch.equal(t_u, t_f)
    li_g, _, t_g = [t.clone() for t in eng.forward_graph(u8)]
    li_g2, _, _ = [t.clone() for t in eng.forward_graph(xf)]        # same batch, other dtype: its own graph
    assert torch.equal(li_g, li_f) and torch.equal(li_g2, li_f)
    lut = input_lut_u8(eng.s0)
    assert lut.shape == (3, 256) and lut.dtype == np.int8 and (np.diff(lut.astype(np.int32), axis=1) >= 0).all()
    eng.set_input_normalisation(mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
    li_o, _, _ = eng.forward(u8)
    xo = ((u8.float().div(255) - 0.5) / 0.5).contiguous()
    assert torch.equal(li_o.clone(), eng.forward(xo)[0])


def test_repeated_forwards_are_identical_at_the_headline_batch():
    """the hand-scheduled kernels issue asynchronous loads behind inline asm; a hazard there would be a rare, data-independent
    difference between two forwards of the same batch.  60 forwards of DeiT-B at batch 256, eager and HIP-graph replay in turn:
    every INT32 logit equals the first forward's (scripts/stress_parity.py runs 400 of each config: profiles/r03zm_*)"""
    fs, ranges, cfg, meta, z = load_synthetic_model("deit_base")
    eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device=DEV, max_batch=256)
    imgs = torch.from_numpy(synth.make_images(16, 77)).to(DEV).repeat(16, 1, 1, 1).contiguous()
    ref = eng.forward(imgs)[0].clone()
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    for it in range(60):
        out = (eng.forward_graph if it % 2 else eng.forward)(imgs)[0]
        bad += (out != ref).any().to(torch.int64)
    assert int(bad) == 0
