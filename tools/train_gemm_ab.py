"""The training step's GEMM shapes (batch 20: M = 5300 rows) on the split kernel's two operand paths: A fp32 split by the
loader + row-major fp16-plane W (what modeling/train.py uses) against packed split planes for both operands staged by
LDS-DMA (what the sampling path uses), per tile configuration.  Decides whether packing the training operands pays.
Run on the GPU box:  python tools/train_gemm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

M0 = 20 * 265
MP = 5376                      # the dW contraction length (M padded to 32 * 8 K-ranges)
SHAPES = [("fwd/dX qkv   ", M0, 3072, 1024, 1), ("fwd/dX proj  ", M0, 1024, 1024, 1), ("fwd fc1/dX fc2", M0, 4096, 1024, 1),
          ("fwd fc2/dX fc1", M0, 1024, 4096, 1), ("dX qkv (K=3072)", M0, 1024, 3072, 1),
          ("dW proj  x8  ", 1024, 1024, MP, 8), ("dW qkv   x4  ", 3072, 1024, MP, 4), ("dW fc1   x4  ", 4096, 1024, MP, 4),
          ("dW fc2   x4  ", 1024, 4096, MP, 4)]


def split(a):
    hi = a.clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half())).contiguous()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {}
for name, M, N, K, S in SHAPES:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    fl = 2.0 * M * N * K
    Wrow = split(W * 2.0 ** 4).view(torch.int16)                     # row-major planes [2][N][K]
    A2 = L.pack_planes(split(A))
    W2p = L.pack_planes(split(W * 2.0 ** 4)).view(torch.int16)
    M16 = (M + 15) // 16 * 16
    out = torch.empty(M, N, device="cuda")
    part = torch.empty(S, M * N, device="cuda")
    row = []
    # (a) the training step's launch: fp32 A, row-major planes, split-K groups for dW
    if S == 1:
        cur = lambda: L.gemm(A, Wrow, out, M, N, K, split2=2.0 ** -4)
    else:
        Kc = K // S
        def cur():
            L.gemm(A, Wrow, part, M, N, Kc, lda=K, ldw=K, ldc=N, groups=S, a_gstride=Kc, w_gstride=Kc, c_gstride=M * N,
                   split2=2.0 ** -4, w_plane=N * K)
            L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(out), 1, S, M * N, M * N, 0, 0, L.stream()))
    t = timeit(cur)
    row.append("loader-split%s %7.1f us %6.1f TF" % (" x%d" % S if S > 1 else "   ", t, fl / t / 1e6))
    tot.setdefault("cur", 0.0)
    tot["cur"] += t
    ref = out.clone()
    # (b) packed operands, LDS-DMA staging, unsplit K, every tile configuration
    best = None
    for tile in (0, 1, 2):
        L.lib().ds_gemm_f16x2_force_tile(tile)
        run = lambda: L.gemm(A2, W2p, out, M, N, K, split2=2.0 ** -4, a_plane=M16 * K)
        t = timeit(run)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        row.append("packed t%d %7.1f us %6.1f TF%s" % (tile, t, fl / t / 1e6, "" if err < 1e-5 else " ERR %.1e" % err))
        best = t if best is None else min(best, t)
    L.lib().ds_gemm_f16x2_force_tile(-1)
    tot.setdefault("packed", 0.0)
    tot["packed"] += best
    print("%s M=%5d N=%4d K=%4d | %s" % (name, M, N, K, " | ".join(row)), flush=True)
print("sum over the listed shapes: loader-split %.0f us, best packed %.0f us" % (tot["cur"], tot["packed"]))
