"""Same-process, interleaved A/B of two builds of the attention kernel on the denoiser's attention-ready path (B = 64, 16
heads; self: 265 keys, cross: 77 keys; 272 query rows = padded-row mode).  usage: attn_ab.py <libA.so> <libB.so>
Prints the median / min launch time per library over interleaved rounds and whether the outputs are bit-identical."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

B, H, D, Lq = 64, 16, 1024, 272
libs = []
for path in sys.argv[1:3]:
    l = C.CDLL(os.path.abspath(path))
    l.ds_attention_f16x2_ready.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_float, C.c_void_p]
    l.ds_attn_nkey.argtypes = [C.c_int]
    libs.append((os.path.basename(path), l))
stream = torch.cuda.current_stream().cuda_stream
for name, Lk in (("self-attention", 265), ("cross-attention", 77)):
    nkey = libs[0][1].ds_attn_nkey(Lk)
    qh = (torch.randn(2, B, H, Lq, 64, device="cuda") * 0.5).half()
    img = (torch.randn(B, H, 4, nkey * 64, device="cuda") * 0.5).half()
    img.view(B, H, 2, 2, -1)            # K hi | K lo | V^T hi | V^T lo
    outs = [torch.zeros(2, (B * Lq + 15) // 16 * 16, D, dtype=torch.float16, device="cuda") for _ in libs]
    times = [[] for _ in libs]

    def run(i):
        rc = libs[i][1].ds_attention_f16x2_ready(qh.data_ptr(), B * H * Lq * 64, img.data_ptr(), outs[i].data_ptr(), D, B, H, Lq,
                                                 Lk, 0.125, stream)
        assert rc == 0
    for i in range(len(libs)):
        for _ in range(5):
            run(i)
    for rnd in range(12):
        for i in range(len(libs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 20 * 1e3)
    for i, (nm, _) in enumerate(libs):
        t = sorted(times[i])
        print("%-16s %-28s median %7.1f us  min %7.1f us" % (name, nm, t[len(t) // 2], t[0]))
    val = [o[0].float() + o[1].float() for o in outs]          # hi + lo, same packed positions in both
    print("%-16s outputs bit-identical: %s, max |a - b| %.3e (max |a| %.3e)"
          % (name, torch.equal(outs[0], outs[1]), float((val[0] - val[1]).abs().max()), float(val[0].abs().max())))
