#!/usr/bin/env python3
"""How reproducible are the reference's OWN calibrated ranges?  (build container only: imports /root/reference)

In calibration mode a QuantAct takes min / max over the float tensor its producer hands it
(/root/reference/models/quantization_utils/quant_modules.py:310-360).  After a QuantLinear that tensor is
`F.linear(x / s, weight_integer, bias_integer) * scale` (:222-226): a float32 sgemm over the NEAR-integers
phi(q) = fl(fl(q*s)/s), whose rounding depends on the BLAS kernel's accumulation order.  This probe calibrates the
reference's DeiT-T twice on the same seeds -- once with 1 thread, once with 8 (a different sgemm blocking) -- and reports
how many of the (x_min, x_max) pairs are bitwise equal between the two runs of the reference itself, and how far apart the
others are.  It also compares both against the fixture tests/golden/deit_tiny_natural.npz (generated with the default
thread count).  Output: tests/golden/calib_variability.json (numbers only)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  (shims + reference import)

rq, ref_models, synth = gg.rq, gg.ref_models, gg.synth


def calibrate(tag, threads):
    torch.set_num_threads(threads)
    factory, wseed, cseeds, cb, iseed, nimg = gg.NATURAL_PLAN[tag]
    model = getattr(ref_models, factory)(pretrained=False, gelu_type="ivit", softmax_type="ivit", layernorm_type="ivit")
    fs = synth.make_float_state(factory, wseed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
    model.eval()
    for cs in cseeds:
        model(torch.from_numpy(synth.make_images(cb, cs)))
    names = [n for n, m in model.named_modules() if isinstance(m, rq.QuantAct)]
    lo = np.array([float(dict(model.named_modules())[n].x_min.item()) for n in names], np.float32)
    hi = np.array([float(dict(model.named_modules())[n].x_max.item()) for n in names], np.float32)
    return names, lo, hi


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def main():
    out = {}
    for tag in ("deit_tiny_natural",):
        z = np.load(os.path.join(gg.GOLD, tag + ".npz"), allow_pickle=True)
        n1, lo1, hi1 = calibrate(tag, 1)
        n8, lo8, hi8 = calibrate(tag, 8)
        assert n1 == n8 == list(z["range_names"])
        d18 = np.maximum(ulps(lo1, lo8), ulps(hi1, hi8))
        d1f = np.maximum(ulps(lo1, z["x_min"]), ulps(hi1, z["x_max"]))
        d8f = np.maximum(ulps(lo8, z["x_min"]), ulps(hi8, z["x_max"]))
        rel18 = float(np.max(np.maximum(np.abs(lo1 - lo8) / np.abs(lo8), np.abs(hi1 - hi8) / np.abs(hi8))))
        first = next((n for n, d in zip(n1, d18) if d), None)
        out[tag] = {"ranges": len(n1), "threads_1_vs_8": {"bitwise_equal": int((d18 == 0).sum()), "max_ulp": int(d18.max()),
                                                           "max_rel": rel18, "first_differing": first},
                    "threads_1_vs_fixture": {"bitwise_equal": int((d1f == 0).sum()), "max_ulp": int(d1f.max())},
                    "threads_8_vs_fixture": {"bitwise_equal": int((d8f == 0).sum()), "max_ulp": int(d8f.max())},
                    "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability()}
        print(tag, json.dumps(out[tag]))
    json.dump(out, open(os.path.join(gg.GOLD, "calib_variability.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
