"""Measure the gather-GEMM on the denoiser's shapes for every block-tile config (HIP events)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_sound_synthesis_amd import _lib

SPLIT = "--bf16x3" in sys.argv
SPLIT2 = "--f16x2" in sys.argv

def bench(M, N, K, tile, act=0, resid=False, iters=10):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
    sc2 = None
    if SPLIT:
        W = _lib.split_bf16x3(W)
    if SPLIT2:
        W, sc2 = _lib.split_f16x2(W)
    b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, device="cuda") if resid else None
    (_lib.lib().ds_gemm_f16x2_force_tile if SPLIT2 else _lib.lib().ds_gemm_bf16x3_force_tile if SPLIT else _lib.lib().ds_gemm_force_tile)(tile)
    for _ in range(2):
        _lib.gemm(A, W, C, M, N, K, bias=b, R=R, act=act, split3=SPLIT, split2=sc2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.gemm(A, W, C, M, N, K, bias=b, R=R, act=act, split3=SPLIT, split2=sc2)
    e1.record(); torch.cuda.synchronize()
    (_lib.lib().ds_gemm_f16x2_force_tile if SPLIT2 else _lib.lib().ds_gemm_bf16x3_force_tile if SPLIT else _lib.lib().ds_gemm_force_tile)(-1)
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    Bs = [int(x) for x in sys.argv[1:] if x.isdigit()] or [32, 64]
    for B in Bs:
        M = B * 265
        for name, N, K, act, res in (("qkv", 3072, 1024, 0, False), ("proj", 1024, 1024, 0, True),
                                     ("fc1", 4096, 1024, 1, False), ("fc2", 1024, 4096, 0, True),
                                     ("logits", 256, 1024, 0, False)):
            row = []
            for tile in (0, 1, 2, -1):
                ms, tf = bench(M, N, K, tile, act, res)
                row.append("%s:%.3fms/%.1fTF" % ({0: "128x128", 1: "128x64", 2: "64x64", -1: "auto"}[tile], ms, tf))
            print("B=%d %-6s M=%d N=%d K=%d  %s" % (B, name, M, N, K, "  ".join(row)), flush=True)
