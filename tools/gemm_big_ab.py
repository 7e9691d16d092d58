"""A/B timing of the packed-operand f16x2 GEMM on the denoiser's shapes (B=64): the per-sample ping-pong program
(gemm_f16x2_ps.hip; the default at this size) against the balanced 128x128 + 64x64-tail launch (force_tile 0), with a
bit-compare of the outputs.  Run on the GPU box:  python tools/gemm_big_ab.py [--batch 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
M = args.batch * 265
SHAPES = [("qkv", 3072, 1024), ("proj/q2", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]
NAMES = {0: "128x128 balanced", 9: "per-sample ping-pong", -1: "default"}


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


total = {t: 0.0 for t in NAMES}
weight = {"qkv": 1, "proj/q2": 3, "fc1": 1, "fc2": 1}       # launches per transformer block
for name, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    R = torch.randn(M, N, device="cuda")
    A2 = L.pack_planes(split(A))
    W2p, sc = L.split_f16x2(W, packed=True)
    M16 = (M + 15) // 16 * 16
    fl = 2.0 * M * N * K
    ref = None
    row = []
    for tile in NAMES:
        L.lib().ds_gemm_f16x2_force_tile(tile)
        out = torch.empty(M, N, device="cuda")
        run = lambda: L.gemm(A2, W2p, out, M, N, K, R=R, split2=sc, a_plane=M16 * K, rows_per_sample=265)
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = torch.equal(out, ref)
        t = timeit(run)
        total[tile] += weight[name] * t
        row.append("%s %7.1f us %6.1f TF%s" % (NAMES[tile], t, fl / t / 1e6, "" if same else " MISMATCH"))
    L.lib().ds_gemm_f16x2_force_tile(-1)
    print("%-8s N=%4d K=%4d | %s" % (name, N, K, " | ".join(row)), flush=True)
print("per block (qkv + 3 x 1024x1024 + fc1 + fc2): " + ", ".join("%s %.0f us" % (NAMES[t], total[t]) for t in NAMES))
