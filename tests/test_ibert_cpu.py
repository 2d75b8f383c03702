"""CPU checks of the I-BERT operator restatement (oracle/ibert.py) against the known-answer vectors produced by the
reference's own modules (tests/golden/ibert_kat.npz) and of the whole-model oracle with that family against the
reference's golden logits (tests/golden/deit_tiny_ibert.npz)."""
import os

import numpy as np
import pytest

from oracle import ibert as ib
from oracle import oracle as orc

import ivit_amd  # noqa: F401
from ivit_amd import synth
from ivit_amd.checkpoint import load_fixture


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "ibert_kat.npz"))


def test_ibert_gelu_kat(kat):
    for i in range(5):
        out, s_out = ib.gelu(kat[f"gelu{i}_k"].astype(np.int32), kat[f"gelu{i}_s"])
        assert np.array_equal(out.astype(np.int32), kat[f"gelu{i}_out"]) and s_out == kat[f"gelu{i}_sout"]
        assert s_out < 0    # the polynomial's leading coefficient is negative: so is the output scale


def test_ibert_softmax_kat(kat):
    for i in range(5):
        lo, hi = kat[f"softmax{i}_range"]
        out, s_out, ninx = ib.softmax(kat[f"softmax{i}_k"].astype(np.int32), kat[f"softmax{i}_s"], lo, hi)
        assert ninx == 0 and s_out == np.float32(2.0 ** -7)
        assert np.array_equal(out.astype(np.int32), kat[f"softmax{i}_out"])
        assert out.min() >= 0 and out.max() <= 128


def test_ibert_layernorm_kat(kat):
    for i in range(4):
        y, s_out, ninx = ib.layernorm(kat[f"ln{i}_k"].astype(np.int32), kat[f"ln{i}_s"], kat[f"ln{i}_gamma"], kat[f"ln{i}_beta"])
        assert ninx == 0 and np.array_equal(s_out, kat[f"ln{i}_sout"])
        got = (y * s_out).astype(np.float32).view(np.int32)
        assert np.array_equal(got, kat[f"ln{i}_out_bits"])      # includes the NaN row of a constant input (std = 0)


def test_oracle_model_with_ibert_family_matches_reference_golden():
    z, meta, ranges = load_fixture("deit_tiny_ibert")
    assert meta["family"] == "ibert" and meta["ln_inexact_rows"] == 0 and meta["softmax_inexact_rows"] == 0
    cfg = synth.MODEL_CONFIGS[meta["factory"]]
    fs = synth.make_float_state(meta["factory"], meta["weight_seed"])
    om = orc.OracleViT(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], family="ibert")
    taps = {}
    res = om.forward(synth.make_images(2, meta["image_seed"]), taps)
    assert np.array_equal(res["logits_int32"], z["logits_int32"][:2])
    assert np.array_equal(res["logits_f32"].view(np.int32), z["logits_f32_bits"][:2])
    assert list(ranges) == synth.qact_names(cfg["depth"], "ibert")
