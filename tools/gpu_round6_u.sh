#!/bin/bash
# Round 6: the one-channel conv_in kernel -- parity tests + its launch time at the training shape (B=20, 80 x 848 -> 128 channels).
O=gpurun_out/${1:-r06u}
mkdir -p $O
python -m pytest tests/test_hip_widening.py tests/test_hip_models.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
python - <<'PY' 2>&1 | tee $O/conv_in_time.txt
import torch
from text_to_sound_synthesis_amd import _lib as L
B, H, W, C = 20, 80, 848, 128
x = torch.rand(B, H, W, device="cuda") * 2 - 1
w = torch.randn(C, 9, device="cuda") * 0.1
b = torch.randn(C, device="cuda") * 0.1
out = torch.empty(B, H, W, C, device="cuda")
part = torch.empty(B, L.lib().ds_conv3x3_c1_chunks(H, W), 2, C, device="cuda", dtype=torch.float64)
def run(p):
    L.check(L.lib().ds_conv3x3_c1(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), B, H, W, C, L.ptr(p) if p is not None else None, L.stream()))
for p in (part, None):
    for _ in range(3): run(p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(p)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print("ds_conv3x3_c1 %s GN partials: %.1f us = %.0f GB/s of output stores" % ("with" if p is not None else "without", t, out.numel() * 4 / t / 1e3))
PY
