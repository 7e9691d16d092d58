"""A few launches of the attention kernel on the denoiser's attention-ready path (B = 64, 16 heads, 272 query rows; self: 265
keys, cross: 77 keys) -- the workload of tools/pmc_sq.sh's counter passes.  DIFFSOUND_LIB selects the library build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

B, H, D, Lq = 64, 16, 1024, 272
lib = L.lib()
for Lk in (265, 77):
    nkey = lib.ds_attn_nkey(Lk)
    qh = (torch.randn(2, B, H, Lq, 64, device="cuda") * 0.5).half()
    img = (torch.randn(B, H, 4, nkey * 64, device="cuda") * 0.5).half()
    out = torch.empty(2, (B * Lq + 15) // 16 * 16, D, dtype=torch.float16, device="cuda")
    for _ in range(6):
        L.check(lib.ds_attention_f16x2_ready(L.ptr(qh), B * H * Lq * 64, L.ptr(img), L.ptr(out), D, B, H, Lq, Lk, 0.125, L.stream()))
    torch.cuda.synchronize()
