import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from text_to_sound_synthesis_amd import synth, _lib as L
from text_to_sound_synthesis_amd.config import build_model, default_config
torch.set_grad_enabled(False)
for nl in (1, 2):
    m = synth.synth_init_(build_model(default_config(n_layer=nl, diffusion_step=100)), seed=0).cuda().eval()
    tr = m.transformer.transformer
    x1 = synth.synth_tokens(1, mask_frac=0.5, key="bs.x").cuda()
    c1 = synth.synth_cond_emb(1, key="bs.c").cuda()
    ref = None
    for B in (1, 8, 15, 16, 17, 20, 21, 33):
        x, c = x1.expand(B, -1).contiguous(), c1.expand(B, -1, -1).contiguous()
        t = torch.full((B,), 41, dtype=torch.long, device="cuda")
        out = tr(x, c, t)
        if ref is None:
            ref = out.clone()
        d = (out - ref.expand(B, -1, -1)).abs()
        bad = (d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
        print("layers %d B=%2d max diff %.3e  differing samples %s" % (nl, B, d.max().item(), bad[:12]), flush=True)
        if bad:
            b = bad[0]
            pos = (d[b].max(0).values > 0).nonzero().flatten()
            print("   sample %d: differing positions %d..%d (%d of 265)" % (b, pos.min().item(), pos.max().item(), pos.numel()))
