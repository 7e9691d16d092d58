"""Two denoiser SAMPLING steps (19 layers, B=64, default f16x2 precision, padded-row mode on) -- the target of the
rocprofv3 --pmc passes that measure the HBM-side traffic of the GEMM launches (profiles/README.md).  It runs
ds_denoiser_step_rng, i.e. exactly what bench.py's timed loop runs per diffusion step, so the counters describe the
instantiations bench.py times (`ds_gemm_f16x2_ps_kernel<*, true>`: 272-row samples) -- round 2 profiled the plain forward,
which never pads rows, and so described the `<*, false>` siblings.  usage: pmc_step.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib, synth
from text_to_sound_synthesis_amd.config import build_model, default_config

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synth.synth_init_(build_model(default_config(n_layer=19, diffusion_step=100)), seed=0).cuda().eval()
dt = model.transformer
dt.truncation_r = 0.85
x = synth.synth_tokens(B, 265, 256, key="pmc.x").cuda()
cond = synth.synth_cond_emb(B, key="pmc.c").cuda()
kv = dt.transformer.condition_kv(cond, dt._schedule_table())
ids = torch.arange(B, device="cuda")
for step in (50, 49):
    t = torch.full((B,), step, dtype=torch.long, device="cuda")
    x = dt.p_sample_tokens_rng(x, kv, t, ids, call=99 - step, initial=False)
torch.cuda.synchronize()
rows = _lib.lib().ds_denoiser_rows_per_sample(dt.transformer.packed(dt._schedule_table())["handle"], B)
print("ok", tuple(x.shape), "rows per sample", rows)
