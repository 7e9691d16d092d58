"""The training step's GEMM shapes (batch 20: M = 5300 rows; cross K|V: 1540 rows) on the packed-operand split kernel
(gemm_f16x2.hip AMODE 2), per tile configuration -- and, for the weight-gradient products dW = dY^T X, per number of K-ranges
S of the split-K launch INCLUDING the fixed-order reduction of the S partial results (ds_colsum).  Decides the dispatch
thresholds of ds_launch_gemm_f16x2 for packed operands and modeling/train.py's _SplitGemm.split_k rule.
Run on the GPU box:  python tools/train_gemm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

M0, MC = 20 * 265, 20 * 77
FWD = [("fwd qkv / dX fc-like N=3072", M0, 3072, 1024), ("fwd proj / q2 / dX proj   ", M0, 1024, 1024),
       ("fwd fc1 / dX fc2          ", M0, 4096, 1024), ("fwd fc2 / dX fc1          ", M0, 1024, 4096),
       ("dX qkv (K = 3072)         ", M0, 1024, 3072), ("fwd kv2 (cond rows)       ", MC, 2048, 512)]
DW = [("dW proj / q2 ", 1024, 1024, M0), ("dW qkv       ", 3072, 1024, M0), ("dW fc1       ", 4096, 1024, M0),
      ("dW fc2       ", 1024, 4096, M0), ("dW kv2       ", 2048, 512, MC)]


def pack(src, rows, cols, rows_pad=0):
    """ds_pack_operand: (row planes, plane stride) if rows_pad == 0 else (transposed planes, plane stride)"""
    if rows_pad == 0:
        pl = (rows + 15) // 16 * 16 * cols
        d = torch.empty(2, pl, dtype=torch.int16, device="cuda")
        L.check(L.lib().ds_pack_operand(L.ptr(src), rows, cols, cols, 1.0, 0, None, 0, L.ptr(d), pl, None, 0, 0, 0, 0, None, None, L.stream()))
    else:
        pl = (cols + 15) // 16 * 16 * rows_pad
        d = torch.empty(2, pl, dtype=torch.int16, device="cuda")
        L.check(L.lib().ds_pack_operand(L.ptr(src), rows, cols, cols, 1.0, 0, None, 0, None, 0, L.ptr(d), pl, rows_pad, 0, 0, None, None, L.stream()))
    return d, pl


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


for name, M, N, K in FWD:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    Ap, apl = pack(A, M, K)
    Wp, wpl = pack(W, N, K)
    out = torch.empty(M, N, device="cuda")
    fl, row, ref = 2.0 * M * N * K, [], None
    for tile in (-1, 0, 1, 2, 3):
        L.lib().ds_gemm_f16x2_force_tile(tile)
        out.fill_(float("nan"))
        t = timeit(lambda: L.gemm(Ap, Wp, out, M, N, K, split2=1.0, a_plane=apl, w_plane=wpl))
        if ref is None:
            ref = out.clone()
        bad = "" if torch.equal(out, ref) else " DIFF %.1e" % float((out - ref).abs().max())
        row.append("%s %6.1f us %5.1f TF%s" % ("auto" if tile < 0 else "t%d" % tile, t, fl / t / 1e6, bad))
    L.lib().ds_gemm_f16x2_force_tile(-1)
    print("%s M=%5d N=%4d K=%4d | %s" % (name, M, N, K, " | ".join(row)), flush=True)

for name, N, K, M in DW:
    dY = torch.randn(M, N, device="cuda")
    X = torch.randn(M, K, device="cuda")
    fl = 2.0 * M * N * K
    dW = torch.empty(N, K, device="cuda")
    best = None
    for S in (1, 2, 3, 4, 5, 6):
        Mp = (M + 32 * S - 1) // (32 * S) * (32 * S)
        a, apl = pack(dY, M, N, Mp)
        w, wpl = pack(X, M, K, Mp)
        part = torch.empty(S, N * K, device="cuda")
        Kc = Mp // S
        row = []
        for tile in (0, 1, 2, 3):
            L.lib().ds_gemm_f16x2_force_tile(tile)

            def run():
                if S == 1:
                    L.gemm(a, w, dW, N, K, Mp, split2=1.0, a_plane=apl, w_plane=wpl)
                else:
                    L.gemm(a, w, part, N, K, Kc, lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16,
                           c_gstride=N * K, split2=1.0, a_plane=apl, w_plane=wpl)
                    L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(dW), 1, S, N * K, N * K, 0, 0, L.stream()))
            t = timeit(run)
            row.append("t%d %6.1f us" % (tile, t))
            if best is None or t < best[0]:
                best = (t, S, tile)
        L.lib().ds_gemm_f16x2_force_tile(-1)
        print("%s N=%4d K=%4d M=%4d S=%d (Mp %4d) | %s" % (name, N, K, M, S, Mp, " | ".join(row)), flush=True)
    print("%s best: %.1f us = %.1f TF-eq at S=%d tile %d" % (name, best[0], fl / best[0] / 1e6, best[1], best[2]), flush=True)

# ---- round 6: the long-contraction forward / dX shapes (336 output tiles of 128 x 128 on 512 slots: a third of the chip idle) as
# split-K launches, INCLUDING the fixed-order reduction of the S partial results -- the price of any stream-K-like scheme on
# these shapes: an output tile is 64 KB of fp32, and every contributor beyond the first writes and re-reads it.
for name, M, N, K in (("fwd fc2 / dX fc1 as split-K", M0, 1024, 4096), ("dX qkv as split-K         ", M0, 1024, 3072),
                      ("fwd proj as split-K       ", M0, 1024, 1024)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    Ap, apl = pack(A, M, K)
    Wp, wpl = pack(W, N, K)
    out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    for S in (1, 2, 3, 4):
        if K % (32 * S):
            continue
        Kc = K // S
        part = torch.empty(S, M * N, device="cuda")
        row = []
        for tile in (0, 1):
            L.lib().ds_gemm_f16x2_force_tile(tile)

            def run():
                if S == 1:
                    L.gemm(Ap, Wp, out, M, N, K, split2=1.0, a_plane=apl, w_plane=wpl)
                else:
                    L.gemm(Ap, Wp, part, M, N, Kc, lda=K, ldw=K, ldc=N, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16,
                           c_gstride=M * N, split2=1.0, a_plane=apl, w_plane=wpl)
                    L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(out), 1, S, M * N, M * N, 0, 0, L.stream()))
            t = timeit(run)
            tg = timeit(lambda: L.gemm(Ap, Wp, part, M, N, Kc, lda=K, ldw=K, ldc=N, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16,
                                       c_gstride=M * N, split2=1.0, a_plane=apl, w_plane=wpl)) if S > 1 else t
            row.append("t%d %6.1f us (GEMM alone %6.1f) %5.1f TF" % (tile, t, tg, fl / t / 1e6))
        L.lib().ds_gemm_f16x2_force_tile(-1)
        print("%s M=%d N=%d K=%d S=%d | %s" % (name, M, N, K, S, " | ".join(row)), flush=True)

# ---- round 6: which tile wins where -- rows M = 265 B for B = 4 .. 48 on the four forward shapes, tiles 0 (128 x 128 / hybrid),
# 1 (128 x 64), 2 (64 x 64), 3 (96 x 128, four waves side by side); '*' marks what the dispatcher takes on its own
if "--batch-sweep" in sys.argv:
    for B in (4, 8, 12, 16, 20, 24, 28, 32, 40, 48):
        M = 265 * B
        for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)):
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda")
            Ap, apl = pack(A, M, K)
            Wp, wpl = pack(W, N, K)
            out = torch.empty(M, N, device="cuda")
            row = []
            ts = {}
            for tile in (-1, 0, 1, 2, 3):
                L.lib().ds_gemm_f16x2_force_tile(tile)
                ts[tile] = timeit(lambda: L.gemm(Ap, Wp, out, M, N, K, split2=1.0, a_plane=apl, w_plane=wpl), n=20)
            L.lib().ds_gemm_f16x2_force_tile(-1)
            best = min((t for t in ts if t >= 0), key=lambda t: ts[t])
            print("B=%2d M=%5d N=%4d K=%4d | auto %6.1f | %s | best t%d %+.0f%% vs auto | t128 %4d t96 %4d" % (
                B, M, N, K, ts[-1], " ".join("t%d %6.1f" % (t, ts[t]) for t in (0, 1, 2, 3)), best,
                100.0 * (ts[best] / ts[-1] - 1.0), -(-M // 128) * -(-N // 128), -(-M // 96) * -(-N // 128)), flush=True)

# ---- round 6: the per-sample programs of the sampling loop (gemm_f16x2_ps.hip: 272 x 256 tiles, 8 waves, one workgroup per CU;
# force_tile 9 = whole-sample tiles, 10 = half tiles) on the training step's 20 samples of 265 rows
if "--per-sample" in sys.argv:
    for name, M, N, K in FWD[:5]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda")
        Ap, apl = pack(A, M, K)
        Wp, wpl = pack(W, N, K)
        out = torch.empty(M, N, device="cuda")
        ref = torch.empty(M, N, device="cuda")
        L.gemm(Ap, Wp, ref, M, N, K, split2=1.0, a_plane=apl, w_plane=wpl)
        fl, row = 2.0 * M * N * K, []
        for tile in (-1, 9, 10):
            L.lib().ds_gemm_f16x2_force_tile(tile)
            out.fill_(float("nan"))
            t = timeit(lambda: L.gemm(Ap, Wp, out, M, N, K, split2=1.0, a_plane=apl, w_plane=wpl, rows_per_sample=265), n=20)
            row.append("%s %6.1f us %5.1f TF%s" % ("auto" if tile < 0 else "t%d" % tile, t, fl / t / 1e6, "" if torch.equal(out, ref) else " DIFF"))
        L.lib().ds_gemm_f16x2_force_tile(-1)
        print("%s M=%5d N=%4d K=%4d | %s" % (name, M, N, K, " | ".join(row)), flush=True)
