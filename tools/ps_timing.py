"""Where the time of the per-sample ping-pong GEMM goes, measured INSIDE the kernel: a probe build of
gemm_f16x2_ps.hip (-DPS_TIMING, tools/build_ps_variant.sh timing -DPS_TIMING -> gpurun_ab_timing.so, loaded through DIFFSOUND_LIB) stamps
s_memrealtime (100 MHz) per workgroup at entry / first operands landed / end of the main loop / after each epilogue slab /
stores drained, plus the shader cycles of prologue + main loop.  Per denoiser GEMM (B = 64) this prints the mean, min
and max of every segment over the workgroups, the launch's span from the first entry to the last drain, and the HIP-event
time of the same launch.  Run on the GPU box:

    bash tools/build_ps_variant.sh timing -DPS_TIMING && DIFFSOUND_LIB=$PWD/gpurun_ab_timing.so python tools/ps_timing.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

B, H = 64, 16
Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 265     # rows per sample: 265, or 272 = the padded-row mode's 16-row ninth block
M = B * Lq
M16 = (M + 15) // 16 * 16
print("rows per sample: %d" % Lq)


def split(a):
    hi = a.clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half())).contiguous()


def ev_time(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [("self/cross proj (row + residual)", 1024, 1024, "row"), ("fc2 (row + residual)", 1024, 4096, "row"),
         ("cross-q (attention-ready Q)", 1024, 1024, "q"), ("qkv (attention-ready Q / K / V^T)", 3072, 1024, "qkv"),
         ("fc1 (GELU2 + packed planes)", 4096, 1024, "split")]
for name, N, K, kind in CASES:
    A2 = L.pack_planes(split(torch.randn(M, K, device="cuda")))
    W2p, sc = L.split_f16x2(torch.randn(N, K, device="cuda") * 0.05, packed=True)
    bias = torch.randn(N, device="cuda")
    tiles = B * N // 256
    tbuf = torch.zeros(tiles * 2 * 10, dtype=torch.int64, device="cuda")
    kw = dict(bias=bias, split2=sc, a_plane=M16 * K, rows_per_sample=Lq, pro_scale=tbuf)
    if kind == "row":
        R = torch.randn(M, N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        run = lambda: L.gemm(A2, W2p, out, M, N, K, R=R, **kw)
    elif kind == "split":
        out = torch.empty(2, M16, N, dtype=torch.float16, device="cuda")
        run = lambda: L.gemm(A2, W2p, out, M, N, K, act=L.ACT_GELU2, c_plane=M16 * N, **kw)
    else:
        qh = torch.empty(2, B, H, Lq, 64, device="cuda", dtype=torch.float16)
        img = torch.zeros(B, H, 4, 288 * 64, device="cuda", dtype=torch.float16) if kind == "qkv" else None
        run = lambda: L.gemm(A2, W2p, qh, M, N, K, store=L.STORE_ATTN, attn=(img, H, 288, B * H * Lq * 64), **kw)
    t_ev = ev_time(run)
    tbuf.zero_()
    run()
    torch.cuda.synchronize()
    t = tbuf.cpu().view(tiles, 2, 10).double()
    if float(t[:, :, 0].max()) == 0.0:
        print("%s: no stamps -- the library was not built with -DPS_TIMING (DIFFSOUND_LIB=%s)" % (name, L.LIB_PATH))
        continue
    ts = t[:, :, :7] * 0.01                                   # microseconds (100 MHz)
    t0 = ts[:, :, 0].min()
    seg = {"prologue": ts[:, :, 1] - ts[:, :, 0], "main loop": ts[:, :, 2] - ts[:, :, 1], "slab 0": ts[:, :, 3] - ts[:, :, 2],
           "slab 1": ts[:, :, 4] - ts[:, :, 3], "slab 2": ts[:, :, 5] - ts[:, :, 4], "drain": ts[:, :, 6] - ts[:, :, 5],
           "whole tile": ts[:, :, 6] - ts[:, :, 0]}
    clk = t[:, :, 7] / ((ts[:, :, 2] - ts[:, :, 0]) * 1e-6) / 1e9
    print("%s  N=%d K=%d: %d tiles (%.2f per CU), HIP events %.1f us, first entry -> last drain %.1f us, last entry at %.1f us, "
          "shader clock %.2f GHz" % (name, N, K, tiles, tiles / 256.0, t_ev, float(ts[:, :, 6].max() - t0),
                                     float(ts[:, :, 0].max() - t0), float(clk.mean())))
    for k, v in seg.items():
        print("    %-10s mean %7.2f  min %7.2f  max %7.2f us   (wave row 0: %7.2f, wave row 1: %7.2f)"
              % (k, float(v.mean()), float(v.min()), float(v.max()), float(v[:, 0].mean()), float(v[:, 1].mean())))
    # order of a CU's tiles: tiles sorted by entry time, in rounds of 256
    order = ts[:, 0, 0].argsort()
    for r in range(0, tiles, 256):
        idx = order[r:r + 256]
        print("    round %d: entry %7.1f .. %7.1f us, end of main loop %7.1f .. %7.1f, drained %7.1f .. %7.1f"
              % (r // 256, float(ts[idx, 0, 0].min() - t0), float(ts[idx, 0, 0].max() - t0), float(ts[idx, 0, 2].min() - t0),
                 float(ts[idx, 0, 2].max() - t0), float(ts[idx, 0, 6].min() - t0), float(ts[idx, 0, 6].max() - t0)))
    sys.stdout.flush()
