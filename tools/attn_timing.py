"""Where a wave of the attention kernel spends its time, measured INSIDE the kernel: a probe build of
attention_f16x2.hip (-DAH_TIMING: tools/build_variant.sh attn_timing attention_f16x2.hip -DAH_TIMING) stamps
s_memrealtime per wave and accumulates, over the key chunks, the time spent waiting for operands / in the scores / at the
barrier + DMA issue / in the online softmax / in P V (round 4: streamed kernel; rounds 2-3 stamped the whole-head phases),
on the denoiser's attention-ready path (B = 64, 16 heads; self: 265 keys, cross: 77 keys).

    DIFFSOUND_LIB=$PWD/gpurun_ab_attn_timing.so python tools/attn_timing.py [rows_per_sample]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import _lib as L

B, H, D = 64, 16, 1024
Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 272
lib = L.lib()
raw = ctypes.CDLL(L.LIB_PATH)
if not hasattr(raw, "ds_attn_timing_buffer"):
    sys.exit("the library was not built with -DAH_TIMING (DIFFSOUND_LIB=%s)" % L.LIB_PATH)
raw.ds_attn_timing_buffer.argtypes = [ctypes.c_void_p]


def ev_time(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, Lk in (("self-attention", 265), ("cross-attention", 77)):
    nkey = lib.ds_attn_nkey(Lk)
    qh = (torch.randn(2, B, H, Lq, 64, device="cuda") * 0.5).half()
    img = (torch.randn(B, H, 4, nkey * 64, device="cuda") * 0.5).half()
    out = torch.empty(2, (B * Lq + 15) // 16 * 16, D, dtype=torch.float16, device="cuda")
    groups = ((Lq + 31) // 32 + 2) // 3
    nwave = B * groups * H * 3
    tbuf = torch.zeros(nwave * 8, dtype=torch.int64, device="cuda")
    raw.ds_attn_timing_buffer(ctypes.c_void_p(tbuf.data_ptr()))
    run = lambda: L.check(lib.ds_attention_f16x2_ready(L.ptr(qh), B * H * Lq * 64, L.ptr(img), L.ptr(out), D, B, H, Lq, Lk, 0.125,
                                                       L.stream()))
    t_ev = ev_time(run)
    tbuf.zero_()
    run()
    torch.cuda.synchronize()
    t = tbuf.cpu().view(nwave, 8).double()
    act = t[:, 7] > 0
    items = t[act][:, 7:8]               # (sample, head, query group) items the wave worked (persistent launch: several)
    ts = t[act][:, :7] * 0.01
    print("%s: %.0f waves resident, %.2f items per wave" % (name, float(act.sum()), float(items.mean())))
    ts[:, 1:6] = ts[:, 1:6] / items      # per item
    # round-4 (streamed) kernel: slot 0 = entry, slots 1..5 = time ACCUMULATED over the chunks in
    # operands landed | scores | barrier + DMA issue | softmax | P V, slot 6 = end
    names = ["wait for K / V^T chunks (+ Q)", "scores S^T = K Q^T", "barrier + next DMA issue", "online softmax", "O^T = V^T P^T"]
    print("%s: Lq %d, Lk %d, %d active waves, HIP events %.1f us, first entry -> last store %.1f us"
          % (name, Lq, Lk, int(act.sum()), t_ev, float(ts[:, 6].max() - ts[:, 0].min())))
    for i, nm in enumerate(names):
        d = ts[:, i + 1]
        print("    %-30s mean %6.2f  min %6.2f  max %6.2f us" % (nm, float(d.mean()), float(d.min()), float(d.max())))
    d = (ts[:, 6] - ts[:, 0]) / items[:, 0]
    rest = d - ts[:, 1:6].sum(1)
    print("    %-30s mean %6.2f  min %6.2f  max %6.2f us" % ("normalise, stage, store", float(rest.mean()), float(rest.min()), float(rest.max())))
    print("    %-30s mean %6.2f  min %6.2f  max %6.2f us" % ("whole item (per wave)", float(d.mean()), float(d.min()), float(d.max())))
    # residency: waves alive per CU-slot ~ sum of lifetimes / span / 256 CUs
    span = float(ts[:, 6].max() - ts[:, 0].min())
    print("    mean waves in flight per CU: %.2f" % (float((ts[:, 6] - ts[:, 0]).sum()) / span / 256.0))
