"""One GEMM shape, one tile config, a few launches -- target for rocprofv3 --pmc passes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_sound_synthesis_amd import _lib
M, N, K, tile = [int(x) for x in sys.argv[1:5]]
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
_lib.lib().ds_gemm_force_tile(tile)
for _ in range(5):
    _lib.gemm(A, W, C, M, N, K, bias=b)
torch.cuda.synchronize()
