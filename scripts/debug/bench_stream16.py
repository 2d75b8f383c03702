import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import ivit_amd as ivit
from ivit_amd import synth
DEV="cuda:0"; B=int(sys.argv[1]) if len(sys.argv)>1 else 256
fam = sys.argv[2] if len(sys.argv) > 2 else "ivit"          # ivit | ibert
pat = sys.argv[3] if len(sys.argv) > 3 else "w16"           # w16 (16-bit stream) | w16all (every width knob at 16)
a16 = 16 if pat == "w16all" else 8
w = dict(patch_embed_bw=16, pos_encoding_bw=a16, block_input_bw=16, attention_out_bw=16, softmax_bw=a16, mlp_out_bw=16, norm2_in_bw=16, att_block_out_bw=16)
fs = synth.make_float_state("deit_base_patch16_224", 7)
model = ivit.deit_base_patch16_224(gelu_type=fam, softmax_type=fam, layernorm_type=fam, **w)
model.load_state_dict({k: torch.from_numpy(v) for k, v in fs.items()}, strict=False)
model.to(DEV).eval()
imgs = torch.from_numpy(synth.make_images(16, 99)).to(DEV).repeat((B+15)//16,1,1,1)[:B].contiguous()
with torch.no_grad():
    model(imgs[:8])
ivit.freeze_model(model)
def timed(n=10):
    with torch.no_grad():
        model(imgs); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): model(imgs)
        torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e3
print(fam, pat, "engine stream16:", model.engine_unsupported_reason(), f"{timed():.2f} ms")
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
    model(imgs); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=60))
model.use_engine=False
print(f"module path: {timed(2):.1f} ms")
