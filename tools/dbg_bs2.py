import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L
torch.set_grad_enabled(False)
torch.manual_seed(0)

def split(a):
    hi = a.clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half())).contiguous()

B, Lq, H, D = 16, 265, 16, 1024
M = B * Lq
M16 = (M + 15) // 16 * 16
A = torch.randn(M, D, device="cuda")
A2p = L.pack_planes(split(A))
for name, N, K, act in (("fc1", 4096, 1024, L.ACT_GELU2), ("qkv", 3072, 1024, L.ACT_NONE), ("proj", 1024, 1024, L.ACT_NONE)):
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    W2, sc = L.split_f16x2(W)
    W2p, _ = L.split_f16x2(W, packed=True)
    ref = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, ref, M, N, K, bias=b, act=act, split2=sc)
    out = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(A2p, W2p, out, M, N, K, bias=b, act=act, split2=sc, a_plane=M16 * K)
    print(name, "row-major packed vs loader-split: equal", torch.equal(out, ref), "nan in ref", torch.isnan(ref).sum().item(), "nan in out", torch.isnan(out).sum().item())
    for nm, tt in (("ref", ref), ("out", out)):
        nn = torch.isnan(tt)
        if nn.any():
            rows = nn.any(1).nonzero().flatten(); cols = nn.any(0).nonzero().flatten()
            print("   ", nm, "nan rows %d..%d (%d)  cols %d..%d (%d)" % (rows.min(), rows.max(), rows.numel(), cols.min(), cols.max(), cols.numel()))
    bad = (out != ref) & ~torch.isnan(out) & ~torch.isnan(ref)
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("    mismatching rows %d..%d (%d) cols %d..%d (%d) maxdiff %.3e" % (rows.min(), rows.max(), rows.numel(), cols.min(), cols.max(), cols.numel(), (out - ref)[bad].abs().max().item()))
    if name == "fc1":
        outs = torch.zeros(2, M16 * N, device="cuda", dtype=torch.float16)
        L.gemm(A2p, W2p, outs, M, N, K, bias=b, act=act, split2=sc, a_plane=M16 * K, c_plane=M16 * N)
        got = L.unpack_planes(outs, M, N)
        want = split(ref)
        badr = (got != want).any(0).any(1).nonzero().flatten()
        print(name, "c_split: equal", torch.equal(got, want), "bad rows", badr[:5].tolist(), "...", badr[-5:].tolist(), badr.numel())
    if name == "qkv":
        heads = lambda x: split(x.contiguous()).view(2, B, Lq, H, 64).permute(0, 1, 3, 2, 4).contiguous()
        q_ref = heads(ref[:, :D]); img_ref = L.attn_images(heads(ref[:, D:2 * D]), heads(ref[:, 2 * D:]), 288)
        qh = torch.full((2, B, H, Lq, 64), float("nan"), device="cuda", dtype=torch.float16)
        img = torch.zeros(B, H, 4, 288 * 64, device="cuda", dtype=torch.float16)
        L.gemm(A2p, W2p, qh, M, N, K, bias=b, split2=sc, a_plane=M16 * K, store=L.STORE_ATTN, rows_per_sample=Lq,
               attn=(img, H, 288, B * H * Lq * 64))
        print(name, "attn store: Q", torch.equal(qh, q_ref), "K", torch.equal(img[:, :, :2], img_ref[:, :, :2]), "VT", torch.equal(img[:, :, 2:], img_ref[:, :, 2:]))
# K=4096 fc2 with residual
K = 4096
A4 = torch.randn(M, K, device="cuda"); A4p = L.pack_planes(split(A4))
W = torch.randn(1024, K, device="cuda") * 0.05; b = torch.randn(1024, device="cuda"); R = torch.randn(M, 1024, device="cuda")
W2, sc = L.split_f16x2(W); W2p, _ = L.split_f16x2(W, packed=True)
ref = torch.empty(M, 1024, device="cuda"); L.gemm(A4, W2, ref, M, 1024, K, bias=b, R=R, split2=sc)
out = torch.empty(M, 1024, device="cuda"); L.gemm(A4p, W2p, out, M, 1024, K, bias=b, R=R, split2=sc, a_plane=M16 * K)
print("fc2 equal", torch.equal(out, ref))
