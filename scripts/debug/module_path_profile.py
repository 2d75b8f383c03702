"""Where the module-by-module path spends its time (host vs device): torch profiler over one frozen forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ivit_amd as ivit
from ivit_amd import synth
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fam = sys.argv[2] if len(sys.argv) > 2 else "ibert"
fs = synth.make_float_state("deit_base_patch16_224", 7)
model = ivit.deit_base_patch16_224(gelu_type=fam, softmax_type=fam, layernorm_type=fam)
model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
model.to(DEV).eval()
imgs = torch.from_numpy(synth.make_images(16, 99)).to(DEV).repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
with torch.no_grad():
    model(imgs[:8])
ivit.freeze_model(model)
model.use_engine = False
with torch.no_grad():
    model(imgs); torch.cuda.synchronize()
    t0 = time.perf_counter(); model(imgs); torch.cuda.synchronize(); print(f"forward {1e3 * (time.perf_counter() - t0):.1f} ms")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model(imgs); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=70))
