#!/usr/bin/env python3
"""Drop-in proof (build container only; needs /root/reference, never shipped to the GPU box).

north_star: the build "drop-in replaces models/layers_quant.py ... behind the same nn.Module API so models/vit_quant.py and
swin_quant.py load unchanged".  This script executes the REFERENCE's own model files -- /root/reference/models/vit_quant.py
and swin_quant.py, byte for byte as they lie there -- with their relative imports `.layers_quant`, `.quantization_utils`
(and `.model_utils`) resolved to the BUILD's package, constructs every factory on CPU, and checks that

  * construction succeeds (every class / function the model files use exists with a compatible signature),
  * the state_dict (keys and shapes) equals the one the reference builds on its own modules
    (tests/golden/state_dict_schema.json, produced by oracle/gen_golden.py schema),
  * load_state_dict of the synthetic weights and freeze_model() / unfreeze_model() run,
  * the frozen reference-file model is recognised by the build's engine builder (same parameter names).

It then records the constructor / forward signatures of the reference's operator classes in
tests/golden/dropin_signatures.json; tests/test_host_logic.py compares the build's classes with that fixture on every run
(no reference needed there).  Harness-side shims only (reference files untouched): a `tkinter` stub for swin_quant.py:2
and a stand-in for `.utils.load_weights_from_npz` (a downloader).
"""
import importlib
import importlib.util
import inspect
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/models"

import torch  # noqa: E402

import ivit_amd  # noqa: E402,F401
import ivit_amd.layers_quant as b_layers  # noqa: E402
import ivit_amd.model_utils as b_mu  # noqa: E402
import ivit_amd.quantization_utils as b_qu  # noqa: E402
from ivit_amd import synth  # noqa: E402


def load_reference_file_on_build_modules(fname):
    """import REF/<fname>.py as a submodule of a synthetic package whose siblings are the build's modules"""
    pkg = types.ModuleType("dropin_models")
    pkg.__path__ = [REF]          # the model files are read from the reference tree, unmodified
    sys.modules["dropin_models"] = pkg
    sys.modules["dropin_models.layers_quant"] = b_layers
    sys.modules["dropin_models.quantization_utils"] = b_qu
    sys.modules["dropin_models.model_utils"] = b_mu
    sys.modules["dropin_models.utils"] = types.SimpleNamespace(load_weights_from_npz=None)   # downloader, unused
    sys.modules.setdefault("tkinter", types.SimpleNamespace(X=None))                          # swin_quant.py:2
    return importlib.import_module(f"dropin_models.{fname}")


def sig(obj):
    out = []
    for name, p in inspect.signature(obj).parameters.items():
        if name in ("self", "args", "kwargs"):
            continue
        d = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append([name, d])
    return out


def reference_signatures():
    """signatures of the reference's own operator classes (imported from the reference package proper)"""
    sys.path.insert(0, "/root/reference")
    torch.Tensor.cuda = lambda self, device=None, *a, **k: self if device is None else self.to(device)
    rq = importlib.import_module("models.quantization_utils")
    rl = importlib.import_module("models.layers_quant")
    rm = importlib.import_module("models.model_utils")
    out = {}
    for mod, names in ((rq, ["QuantLinear", "QuantAct", "QuantMatMul", "QuantConv2d", "IVITIntLayerNorm", "IVITIntGELU",
                             "IVITIntSoftmax", "IBERTIntLayerNorm", "IBERTIntGELU", "IBERTIntSoftmax"]),
                       (rl, ["Mlp", "PatchEmbed", "DropPath"])):
        for n in names:
            cls = getattr(mod, n)
            out[n] = {"init": sig(cls.__init__), "forward": sig(cls.forward),
                      "methods": sorted(m for m in ("fix", "unfix") if callable(getattr(cls, m, None)))}
    for n in ("get_gelu", "get_softmax", "get_layernorm", "attach_io_stat_hooks", "save_io_stats_df", "enable_io_stats",
              "disable_io_stats", "clear_io_stats", "softmax"):      # the last six: the I/O-statistics collector + a float helper
        out[n] = {"call": sig(getattr(rq, n))}
    for n in ("freeze_model", "unfreeze_model"):
        out[n] = {"call": sig(getattr(rm, n))}
    ru = importlib.import_module("models.quantization_utils.quant_utils")     # the functional layer under the modules
    for n in ("linear_quantize", "symmetric_linear_quantization_params", "batch_frexp"):
        out[n] = {"call": sig(getattr(ru, n))}
    for n in ("SymmetricQuantFunction", "floor_ste", "round_ste", "fixedpoint_mul"):
        out[n] = {"autograd_forward": sig(getattr(ru, n).forward)}
    return out


def main():
    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")))
    report = {}
    vq = load_reference_file_on_build_modules("vit_quant")
    assert vq.__file__.startswith(REF), vq.__file__
    for factory in ("deit_tiny_patch16_224", "deit_small_patch16_224", "deit_base_patch16_224"):
        model = getattr(vq, factory)(pretrained=False, gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit")
        sd = {k: list(v.shape) for k, v in model.state_dict().items()}
        assert sd == schema[factory], f"{factory}: state_dict of the reference file on the build's modules differs"
        fs = synth.make_float_state(factory, 11)
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
        assert not unexpected
        b_mu.freeze_model(model)
        assert all(not m.running_stat for m in model.modules() if isinstance(m, b_qu.QuantAct))
        b_mu.unfreeze_model(model)
        assert all(m.running_stat for m in model.modules() if isinstance(m, b_qu.QuantAct))
        report[factory] = {"keys": len(sd), "operator_classes": sorted({type(m).__name__ for m in model.modules()
                                                                       if type(m).__module__.startswith("i-vit_amd")})}
    # the fork's default family too (vit_quant.py:188-190)
    model = vq.deit_tiny_patch16_224(pretrained=False)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == schema["deit_tiny_patch16_224@ibert"]
    report["deit_tiny_patch16_224@ibert"] = {"keys": len(model.state_dict())}
    # 16-bit width knobs flow through the reference's constructor into the build's QuantActs
    model = vq.deit_tiny_patch16_224(pretrained=False, gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit",
                                     att_block_out_bw=16, norm2_in_bw=16)
    assert model.blocks[0].qact4.activation_bit == 16 and model.blocks[0].qact2.activation_bit == 16

    sq = load_reference_file_on_build_modules("swin_quant")
    assert sq.__file__.startswith(REF), sq.__file__
    model = sq.swin_tiny_patch4_window7_224(pretrained=False)
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sd == schema["swin_tiny_patch4_window7_224"], "swin: state_dict differs"
    fs = synth.make_swin_float_state("swin_tiny_patch4_window7_224", 21)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    assert not unexpected
    b_mu.freeze_model(model)
    report["swin_tiny_patch4_window7_224"] = {"keys": len(sd)}

    sigs = reference_signatures()
    out = {"reference_files_on_build_modules": report, "signatures": sigs}
    with open(os.path.join(ROOT, "tests", "golden", "dropin_signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("drop-in proof ok:", json.dumps(report))


if __name__ == "__main__":
    main()
