"""One GEMM shape, one tile config, a few launches -- target for rocprofv3 --pmc passes.
usage: pmc_gemm.py M N K tile [fp32|f16x2]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_sound_synthesis_amd import _lib
M, N, K, tile = [int(x) for x in sys.argv[1:5]]
mode = sys.argv[5] if len(sys.argv) > 5 else "fp32"
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
kw = {}
if mode == "f16x2":
    W, sc = _lib.split_f16x2(W); kw = dict(split2=sc); _lib.lib().ds_gemm_f16x2_force_tile(tile)
else:
    _lib.lib().ds_gemm_force_tile(tile)
for _ in range(5):
    _lib.gemm(A, W, C, M, N, K, bias=b, **kw)
torch.cuda.synchronize()
