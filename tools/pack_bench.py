"""ds_pack_operand alone on the training step's shapes (row + transposed forms, fp32 source): bytes per second.
Run on the GPU box:  [DIFFSOUND_LIB=...] python tools/pack_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

for rows, cols in ((5300, 1024), (5300, 3072), (5300, 4096)):
    src = torch.randn(rows, cols, device="cuda")
    Mp = (rows + 31) // 32 * 32
    R16 = (rows + 15) // 16 * 16
    row = torch.empty(2, R16 * cols, dtype=torch.int16, device="cuda")
    t = torch.empty(2, cols * Mp, dtype=torch.int16, device="cuda")
    part = torch.empty(L.lib().ds_pack_operand_tile_rows(rows, Mp), cols, device="cuda")

    def run():
        L.check(L.lib().ds_pack_operand(L.ptr(src), rows, cols, cols, 1.0, 0, None, 0, L.ptr(row), R16 * cols, L.ptr(t), cols * Mp, Mp, 0, 0,
                                        L.ptr(part), None, L.stream()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%s: %d x %d: %.1f us = %.2f TB/s (12 B per element)" % (os.environ.get("DIFFSOUND_LIB", "product")[-24:], rows, cols, us, rows * cols * 12 / us / 1e6))
