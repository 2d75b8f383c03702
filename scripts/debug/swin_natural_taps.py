"""Which tap of the Swin module path first differs from the reference at natural scales? (GPU box)"""
import os, sys, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ivit_amd as ivit
import ivit_amd.quantization_utils as qu
from ivit_amd import synth
from ivit_amd.checkpoint import load_fixture
DEV = "cuda:0"
z, meta, ranges = load_fixture("swin_tiny_natural")
fs = synth.make_swin_float_state(meta["factory"], meta["weight_seed"])
model = ivit.swin_tiny_patch4_window7_224()
model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
for name, mod in model.named_modules():
    if isinstance(mod, qu.QuantAct) and name in ranges:
        mod.x_min.fill_(float(ranges[name][0])); mod.x_max.fill_(float(ranges[name][1]))
model.to(DEV); ivit.freeze_model(model)
gold = dict(zip([str(x) for x in z["tap_names"]], z["tap_crc32"]))
order, got = [], {}
def hook(name):
    def fn(mod, inp, outp):
        y, s = outp
        v = torch.round(y / s.reshape(-1)[0] if s.numel() == 1 else y / s).to(torch.int64).cpu().numpy().astype(np.int32)
        got[name] = zlib.crc32(np.ascontiguousarray(v).tobytes()); order.append(name)
    return fn
for name, mod in model.named_modules():
    if isinstance(mod, (qu.QuantAct, qu.IVITIntSoftmax, qu.IVITIntGELU)) and name != "act_out":
        mod.register_forward_hook(hook(name))
imgs = torch.from_numpy(synth.make_images(meta["n_images"], meta["image_seed"])).to(DEV)
with torch.no_grad():
    model(imgs)
bad = [n for n in order if n in gold and got[n] != int(gold[n])]
print("taps", len(order), "bad", len(bad)); print("first bad:", bad[:8]); print("order head:", order[:12])
