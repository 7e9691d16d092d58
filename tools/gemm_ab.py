"""A/B timing of the f16x2 GEMM staging modes on the denoiser's shapes (B=64): fp32 A split in the loader,
packed split planes staged by LDS-DMA.  Run on the GPU box: python tools/gemm_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

M = 64 * 265
SHAPES = [("qkv", 3072, 1024), ("proj/q2", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("logits", 256, 1024)]


def split(a):
    hi = a.clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half())).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    W2, sc = L.split_f16x2(W)
    A2 = L.pack_planes(split(A))
    W2p, _ = L.split_f16x2(W, packed=True)
    out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    for tile in (0, 1):
        L.lib().ds_gemm_f16x2_force_tile(tile)
        row = []
        t = timeit(lambda: L.gemm(A, W2, out, M, N, K, split2=sc))
        row.append("loader-split %7.1f us %6.1f TF" % (t, fl / t / 1e6))
        t = timeit(lambda: L.gemm(A2, W2p, out, M, N, K, split2=sc, a_plane=M * K))
        row.append("packed-dma %7.1f us %6.1f TF" % (t, fl / t / 1e6))
        print("%-8s N=%4d K=%4d tile %s | %s" % (name, N, K, ("128x128", "128x64")[tile], " | ".join(row)), flush=True)
    L.lib().ds_gemm_f16x2_force_tile(-1)
