"""Which event flavour times ONE kernel inside the DeiT-B b256 forward without perturbing it?  Compares event pairs
around every fused-residual GEMM (torch.cuda.Event vs raw HIP events with different release scopes) with each other;
ground truth = rocprofv3 --kernel-trace --stats of an unprobed bench run (profiles/).  Also reports the forward's
wall time under each probe (how much the probe costs)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ivit_amd  # noqa
from ivit_amd import synth, hiptime
from ivit_amd.checkpoint import load_synthetic_model
from ivit_amd.engine import IntViTEngine

fs, ranges, cfg, meta, _ = load_synthetic_model("deit_base")
eng = IntViTEngine(fs, ranges, cfg["embed_dim"], cfg["depth"], cfg["num_heads"], device="cuda:0", max_batch=256)
images = torch.from_numpy(synth.make_images(256, 5000)).to("cuda:0")
for _ in range(5):
    eng.forward(images)
torch.cuda.synchronize()


class TorchProbe:
    def __init__(self): self.rows, self._o = [], None
    def begin(self, tag, st):
        e = torch.cuda.Event(enable_timing=True); e.record(); self._o = e
    def end(self, tag, st, work):
        e = torch.cuda.Event(enable_timing=True); e.record(); self.rows.append((tag, self._o, e, work))
    def results(self):
        return [(t, a.elapsed_time(b), w) for t, a, b, w in self.rows]


def wall(n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.forward(images)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

out = {"forward_ms_unprobed": wall(), "hip_runtime": hiptime.hip()._name}
modes = {"torch": None, "hip_default": hiptime.hipEventDefault, "hip_release_device": hiptime.hipEventReleaseToDevice,
         "hip_nosysfence": hiptime.hipEventDisableSystemFence, "hip_release_system": 0x80000000}
for name, flags in modes.items():
    p = TorchProbe() if flags is None else hiptime.KernelProbe(flags=flags)
    eng.probe = p
    try:
        w = wall(3)
        eng.probe = None
        rows = p.results(); out.setdefault("empty_pair_us", {})[name] = round(1e3*sum(r[3] for r in rows)/len(rows),2) if len(rows[0])>3 else None
    except RuntimeError as ex:
        eng.probe = None
        torch.cuda.synchronize()
        out[name] = str(ex)
        continue
    k768 = [ms for _, ms, (M, N, K) in rows if K == 768]
    k3072 = [ms for _, ms, (M, N, K) in rows if K == 3072]
    out[name] = {"forward_ms": round(w, 3), "avg_us_all": round(1e3 * sum(r[1] for r in rows) / len(rows), 1),
                 "avg_us_proj": round(1e3 * sum(k768) / len(k768), 1), "avg_us_fc2": round(1e3 * sum(k3072) / len(k3072), 1)}
out["forward_ms_unprobed_after"] = wall()
print(json.dumps(out, indent=1))
