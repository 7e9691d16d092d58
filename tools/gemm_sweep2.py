"""Interleaved, repeated timing of the denoiser GEMM shapes per block-tile config (median of rounds)."""
import sys, os, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_sound_synthesis_amd import _lib
MODE = "f16x2" if "--f16x2" in sys.argv else "bf16x3" if "--bf16x3" in sys.argv else "fp32"
force = {"f16x2": _lib.lib().ds_gemm_f16x2_force_tile, "bf16x3": _lib.lib().ds_gemm_bf16x3_force_tile,
         "fp32": _lib.lib().ds_gemm_force_tile}[MODE]

def run(B):
    M = B * 265
    for name, N, K, act, res in (("qkv", 3072, 1024, 0, False), ("proj", 1024, 1024, 0, True),
                                 ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True), ("logits", 256, 1024, 0, False)):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
        b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, device="cuda") if res else None
        kw = {}
        if MODE == "f16x2":
            W, sc = _lib.split_f16x2(W); kw = dict(split2=sc)
        elif MODE == "bf16x3":
            W = _lib.split_bf16x3(W); kw = dict(split3=True)
        times = {0: [], 1: [], 2: []}
        for rnd in range(7):
            for tile in (0, 1, 2):
                force(tile)
                _lib.gemm(A, W, C, M, N, K, bias=b, R=R, act=act, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    _lib.gemm(A, W, C, M, N, K, bias=b, R=R, act=act, **kw)
                e1.record(); torch.cuda.synchronize()
                if rnd >= 2:
                    times[tile].append(e0.elapsed_time(e1) / 6)
        force(-1)
        row = ["%s:%.3fms/%.0fTF" % (nm, statistics.median(times[t]), 2.0 * M * N * K / statistics.median(times[t]) / 1e9)
               for t, nm in ((0, "128x128"), (1, "128x64"), (2, "64x64"))]
        print("%s B=%d %-6s N=%d K=%d  %s" % (MODE, B, name, N, K, "  ".join(row)), flush=True)

for B in [int(x) for x in sys.argv[1:] if x.isdigit()] or [64]:
    run(B)
