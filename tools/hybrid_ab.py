"""Balanced hybrid launch vs plain 128x128 grid on the denoiser's shapes (B=64).  python tools/hybrid_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L
M = 64 * 265
M16 = (M + 15) // 16 * 16


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


for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    A2p = L.pack_planes(split(A)); W2p, sc = L.split_f16x2(W, packed=True)
    out = torch.empty(M, N, device="cuda")
    row = []
    for label, tile, slots in (("plain 128x128", 0, 1 << 30), ("hybrid", 0, 512), ("128x64", 1, 512)):
        L.lib().ds_gemm_f16x2_force_tile(tile); L.lib().ds_gemm_f16x2_set_balance_slots(slots)
        t = timeit(lambda: L.gemm(A2p, W2p, out, M, N, K, bias=b, split2=sc, a_plane=M16 * K))
        row.append("%s %6.1f us %5.1f TF" % (label, t, 2.0 * M * N * K / t / 1e6))
    L.lib().ds_gemm_f16x2_force_tile(-1); L.lib().ds_gemm_f16x2_set_balance_slots(512)
    print("%-5s N=%4d K=%4d | %s" % (name, N, K, " | ".join(row)), flush=True)
