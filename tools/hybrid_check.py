"""Correctness sweep of the packed-operand f16x2 GEMM launch variants (plain tiles, balanced hybrid with several
balance units) against float64 on grids of many rounds -- the configuration in which a missing LDS-DMA wait once
corrupted results.  python tools/hybrid_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L
torch.set_grad_enabled(False)
torch.manual_seed(0)

def split(a):
    hi = a.clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half())).contiguous()

def report(name, out, ref64):
    nn = torch.isnan(out)
    err = ((out.double() - ref64).abs() / ref64.abs().max()).masked_fill(nn, 0)
    bad = (err > 1e-5) | nn
    msg = "%-34s nan %8d  bad %8d" % (name, nn.sum().item(), bad.sum().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        msg += "  rows %d..%d (%d) cols %d..%d (%d)" % (rows.min(), rows.max(), rows.numel(), cols.min(), cols.max(), cols.numel())
    print(msg, flush=True)

for M, N, K in ((4240, 4096, 1024), (4240, 2048, 1024), (1100, 4096, 1024), (4240, 1536, 1024)):
    M16 = (M + 15) // 16 * 16
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    ref64 = A.double() @ W.double().t() + b.double()
    A2p = L.pack_planes(split(A)); W2, sc = L.split_f16x2(W); W2p, _ = L.split_f16x2(W, packed=True)
    print("M=%d N=%d K=%d" % (M, N, K))
    out = torch.full((M, N), float("nan"), device="cuda"); L.gemm(A, W2, out, M, N, K, bias=b, split2=sc); report("loader-split auto", out, ref64)
    for tile, slots, label in ((1, 512, "packed 128x64"), (2, 512, "packed 64x64"), (0, 1 << 30, "packed 128x128 plain"),
                               (0, 512, "packed hybrid slots 512"), (0, 64, "packed hybrid slots 64"), (0, 1, "packed hybrid slots 1")):
        L.lib().ds_gemm_f16x2_force_tile(tile); L.lib().ds_gemm_f16x2_set_balance_slots(slots)
        out = torch.full((M, N), float("nan"), device="cuda")
        L.gemm(A2p, W2p, out, M, N, K, bias=b, split2=sc, a_plane=M16 * K)
        report(label, out, ref64)
    L.lib().ds_gemm_f16x2_force_tile(-1); L.lib().ds_gemm_f16x2_set_balance_slots(512)
