"""GPU calibration of the module mirror on the fixtures' calibration seeds against the reference's calibrated ranges
(tests/golden/*_natural.npz: x_min / x_max of every QuantAct after the reference's own calibration forwards).
Prints, per fixture, how many ranges are bitwise equal, the largest deviation in ulps and where the first one is."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivit_amd as ivit  # noqa: E402
import ivit_amd.quantization_utils as qu  # noqa: E402
from ivit_amd import synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
DEV = "cuda:0"


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


for tag in sys.argv[1:] or ["deit_tiny_natural", "deit_small_natural", "deit_tiny_ibert_natural"]:
    z = np.load(os.path.join(GOLD, tag + ".npz"), allow_pickle=True)
    meta = json.loads(str(z["meta"]))
    fam = meta.get("family", "ivit")
    model = getattr(ivit, meta["factory"])(gelu_type=fam, softmax_type=fam, layernorm_type=fam)
    fs = synth.make_float_state(meta["factory"], meta["weight_seed"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        for cs in meta["calib_seeds"]:
            model(torch.from_numpy(synth.make_images(meta["calib_batch"], cs)).to(DEV))
    mods = dict(model.named_modules())
    names = [str(n) for n in z["range_names"]]
    lo = np.array([float(mods[n].x_min) for n in names], np.float32)
    hi = np.array([float(mods[n].x_max) for n in names], np.float32)
    d = np.maximum(ulps(lo, z["x_min"]), ulps(hi, z["x_max"]))
    rel = np.maximum(np.abs(lo - z["x_min"]) / np.maximum(np.abs(z["x_min"]), 1e-30), np.abs(hi - z["x_max"]) / np.maximum(np.abs(z["x_max"]), 1e-30))
    first = next((n for n, dd in zip(names, d) if dd), None)
    print(f"{tag}: {len(names)} ranges, bitwise equal {(d == 0).sum()}, max {d.max()} ulp, max rel {rel.max():.3e}, first differing {first}")
    for n, dd, a, b, c, e in zip(names, d, lo, z["x_min"], hi, z["x_max"]):
        if dd:
            print(f"   {n:40s} {dd:6d} ulp   min {a!r} vs {b!r}   max {c!r} vs {e!r}")
