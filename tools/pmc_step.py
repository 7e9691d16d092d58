"""One denoiser forward step (19 layers, B=64, default f16x2 precision) -- target for rocprofv3 --pmc passes that
measure the HBM-side traffic of the GEMM launches (profiles/README.md).  usage: pmc_step.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import synth
from text_to_sound_synthesis_amd.config import build_model, default_config

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synth.synth_init_(build_model(default_config(n_layer=19, diffusion_step=100)), seed=0).cuda().eval()
tr = model.transformer.transformer
x = synth.synth_tokens(B, 265, 256, key="pmc.x").cuda()
cond = synth.synth_cond_emb(B, key="pmc.c").cuda()
t = torch.full((B,), 50, dtype=torch.long, device="cuda")
for _ in range(2):
    out = tr(x, cond, t)
torch.cuda.synchronize()
print("ok", tuple(out.shape))
