"""Time ds_conv3x3_f16x2 on the decoder's three hot geometries at B = 64 (HIP events; TF-eq = algorithmic 2 M N K flops / time).
DIFFSOUND_LIB selects the library build.  python tools/conv3x3_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

lib = L.lib()
B = 64
for H, W, Cin, Cout, gn, up in ((80, 848, 128, 128, True, 0), (40, 424, 256, 128, True, 0), (20, 212, 256, 256, True, 0),
                                (80, 848, 128, 128, False, 1)):
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, hs, ws, Cin, device="cuda")
    w = torch.randn(Cout, 9 * Cin, device="cuda") * 0.05
    w2, sc = L.split_f16x2(w)
    wq = L.pack_conv3x3_weights(w2, Cout, Cin)
    bias = torch.randn(Cout, device="cuda")
    ps, po = (torch.rand(B, Cin, device="cuda") + 0.5, torch.randn(B, Cin, device="cuda")) if gn else (None, None)
    y = torch.empty(B, H, W, Cout, device="cuda")
    part = torch.empty(B, lib.ds_conv3x3_tiles(H, W), 2, Cout, device="cuda", dtype=torch.float64)
    run = lambda: L.check(lib.ds_conv3x3_f16x2(L.ptr(x), L.ptr(wq), wq.numel(), sc, L.ptr(bias), None, L.ptr(y), B, H, W, Cin, Cout, up,
                                               L.ptr(ps), L.ptr(po), L.ptr(part), L.stream()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print("conv3x3 %dx%d %d->%d %s: %.3f ms  %.0f TF-eq (%.2f of the 3-pass ceiling 833)"
          % (H, W, Cin, Cout, "gn+swish" if gn else ("upsampled source" if up else "plain"), ms, fl / ms / 1e9, fl / ms / 1e9 / 833.3))
