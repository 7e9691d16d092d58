"""The halo-tiled 3x3 convolution (csrc/conv3x3_f16x2.hip: ds_conv3x3_f16x2, the SpecVQGAN decoder's hot convs --
specvqgan/modules/diffusionmodules/model.py:92-151 ResnetBlock, :37-52 Upsample) against float64 and against the
tap-by-tap kernel it replaces, and the GroupNorm partial sums of its epilogue (ds_groupnorm_finish) against the
statistics pass (ds_groupnorm_stats).  GPU only; everything through libdiffsound_hip.so."""
import pytest
import torch
import torch.nn.functional as F

from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
NO_GRAD = True


@pytest.fixture(scope="module")
def L():
    from text_to_sound_synthesis_amd import _lib
    _lib.lib()
    return _lib


def rnd(shape, key, scale=1.0):
    return (synth.synth_uniform(shape, key=key) * 2 - 1) * scale


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30)).item()


# (B, H, W, Cin, Cout): whole tiles, ragged tiles in both directions (H % 4, W % 32 != 0), several slabs / n-tiles
SHAPES = [(2, 20, 212, 64, 128), (1, 6, 40, 32, 128), (2, 8, 64, 128, 256), (1, 40, 424, 256, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "gn", "up"])
def test_conv3x3_halo_vs_float64_and_gather_kernel(L, shape, mode):
    B, H, W, Cin, Cout = shape
    up = 1 if mode == "up" else 0
    gn = mode == "gn"
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = rnd((B, hs, ws, Cin), "c3.x", 2.0)
    x[0, 0, 0, :4] = torch.tensor([40.0, -35.0, 1e-4, 0.0])           # a corner with large values: padding / halo edges
    w, bias = rnd((Cout, Cin, 3, 3), "c3.w", 0.1), rnd((Cout,), "c3.b")
    R = rnd((B, H, W, Cout), "c3.r")
    xin = x.permute(0, 3, 1, 2).double()
    sc = sh = None
    if gn:
        sc, sh = rnd((B, Cin), "c3.s") + 1.5, rnd((B, Cin), "c3.o")
        xin = xin * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = (F.conv2d(xin, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + R.double())
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().cuda()
    w2, scale = L.split_f16x2(wp)
    xc, Rc, bc = x.cuda(), R.cuda(), bias.cuda()
    scc, shc = (sc.cuda(), sh.cuda()) if gn else (None, None)
    lib = L.lib()
    tiles = lib.ds_conv3x3_tiles(H, W)
    assert tiles == ((H + 3) // 4) * ((W + 31) // 32)
    part = torch.full((B, tiles, 2, Cout), float("nan"), device="cuda", dtype=torch.float64)
    y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    wq = L.pack_conv3x3_weights(w2, Cout, Cin)
    L.check(lib.ds_conv3x3_f16x2(L.ptr(xc), L.ptr(wq), wq.numel(), scale, L.ptr(bc), L.ptr(Rc), L.ptr(y), B, H, W, Cin, Cout,
                                 up, L.ptr(scc), L.ptr(shc), L.ptr(part), L.stream()))
    old = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    L.gemm(xc, w2, old, B * H * W, Cout, 9 * Cin, split2=scale, conv_split=True, bias=bc, R=Rc, loader=L.LOAD_CONV2D,
           pro=L.PRO_AFFINE_SWISH if gn else L.PRO_NONE, pro_scale=scc, pro_shift=shc, Cin=Cin, H=H, Wd=W, up=up)
    e_new, e_old = relerr(y.cpu(), ref), relerr(old.cpu(), ref)
    print("conv3x3 %s %s: halo %.2e, gather kernel %.2e (relative max error vs float64)" % (shape, mode, e_new, e_old))
    assert torch.isfinite(y).all()
    assert e_new < max(3e-6, 1.2 * e_old)
    # the epilogue's partial sums are the sums of the STORED values (double accumulation)
    yd = y.double().cpu()
    s_ref, q_ref = yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))
    pc = part.cpu()
    assert torch.isfinite(pc).all()
    assert (pc[:, :, 0].sum(1) - s_ref).abs().max() <= 1e-9 * max(1.0, float(q_ref.max()))
    assert (pc[:, :, 1].sum(1) - q_ref).abs().max() <= 1e-9 * float(q_ref.max())
    # ... and ds_groupnorm_finish on them gives the affine of the statistics pass over y
    if Cout % 32 == 0:
        gam, bet = rnd((Cout,), "c3.g") + 1.0, rnd((Cout,), "c3.be")
        a0, b0 = torch.empty(B, Cout, device="cuda"), torch.empty(B, Cout, device="cuda")
        a1, b1 = torch.empty(B, Cout, device="cuda"), torch.empty(B, Cout, device="cuda")
        P = H * W
        work = torch.empty(B * ((P + 255) // 256) * 2 * Cout, device="cuda", dtype=torch.float64)
        gc, bec = gam.cuda(), bet.cuda()
        L.check(lib.ds_groupnorm_stats(L.ptr(y), B, P, Cout, 32, L.ptr(gc), L.ptr(bec), 1e-6, L.ptr(work), L.ptr(a0),
                                       L.ptr(b0), L.stream()))
        L.check(lib.ds_groupnorm_finish(L.ptr(part), B, tiles, P, Cout, 32, L.ptr(gc), L.ptr(bec), 1e-6, L.ptr(a1),
                                        L.ptr(b1), L.stream()))
        assert (a0 - a1).abs().max().item() <= 2e-6 * a0.abs().max().item()
        assert (b0 - b1).abs().max().item() <= 2e-6 * max(1.0, b0.abs().max().item())


def test_decoder_same_mel_in_both_arithmetic_modes():
    """VQModel.decode in the default mode (halo-tiled split convs + folded GroupNorm statistics) against the strict mode
    (every conv a gather-GEMM on the exact-fp32 MFMA): the mel agrees far inside the north-star tolerance."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=1))
    m.load_state_dict(dict(synth_sd("dalle", 1)), strict=False)
    m = m.cuda().eval()
    tok = synth.synth_tokens(2, mask_frac=0.0, key="c3.codes").cuda()
    codec = m.content_codec
    assert codec.conv_precision == "f16x2"
    a = m.decode_to_img(tok, (2, 256, 5, 53)).cpu()
    codec.conv_precision = "fp32"
    try:
        b = m.decode_to_img(tok, (2, 256, 5, 53)).cpu()
    finally:
        codec.conv_precision = "f16x2"
    d = (a - b).abs().max().item()
    print("decode: f16x2 halo-tiled convs vs exact-fp32 gather convs, mel max-abs difference %.2e (mel range %.2f)" % (d, float(b.abs().max())))
    assert torch.isfinite(a).all() and d < 1e-4


def test_melgan_resblock_tail_one_gemm():
    """ds_melgan_resblock_tail (vocoder/modules.py:72-85: 1x1 conv on the activated k3 output + 1x1 shortcut as ONE contraction
    over [LReLU(h) | x]) against float64."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd import _lib as L
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    M, C = 1000, 64
    h, x = rnd((M, C), "rt.h", 3.0), rnd((M, C), "rt.x", 3.0)
    w2, ws = rnd((C, C), "rt.w2", 0.2), rnd((C, C), "rt.ws", 0.2)
    b = rnd((C,), "rt.b")
    ref = (F.leaky_relu(h.double(), 0.2) @ w2.double().t() + x.double() @ ws.double().t() + b.double())
    planes, osc = L.split_f16x2(torch.cat((w2, ws), 1).contiguous().cuda())
    y = torch.full((M, C), float("nan"), device="cuda")
    hc, xc, bc = h.cuda(), x.cuda(), b.cuda()          # (kept alive: a temporary's memory is reused by the next allocation)
    L.check(L.lib().ds_melgan_resblock_tail(L.ptr(hc), L.ptr(xc), L.ptr(planes), C * 2 * C, osc, L.ptr(bc), L.ptr(y), M, C,
                                            L.stream()))
    assert relerr(y.cpu(), ref) < 3e-6


def _resblock_ref64(x, w3, b3, w2, b2, ws, bs, dil):
    """vocoder/modules.py:72-85 in float64: x [B][T][C] channels-last; w3 [C][C][3], w2 / ws [C][C]."""
    xc = x.double().permute(0, 2, 1)                                           # [B][C][T]
    h = F.conv1d(F.pad(F.leaky_relu(xc, 0.2), (dil, dil), mode="reflect"), w3.double(), b3.double(), dilation=dil)
    y = F.conv1d(F.leaky_relu(h, 0.2), w2.double()[:, :, None], b2.double()) + F.conv1d(xc, ws.double()[:, :, None], bs.double())
    return y.permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("dil", [1, 3, 9])
@pytest.mark.parametrize("C,T", [(32, 640), (64, 768)])
def test_melgan_resblock_single_pass(C, T, dil):
    """ds_melgan_resblock(h = NULL): the 32- / 64-channel ResnetBlock as ONE kernel (melgan_fused.hip) against float64 and against
    the two-launch form, clip ends (reflection) and tile borders included; three clips so that a persistent workgroup crosses clips."""
    from text_to_sound_synthesis_amd import _lib as L
    B = 3
    assert L.lib().ds_melgan_resblock_fused_ok(T, C, dil) == 1
    assert L.lib().ds_melgan_resblock_fused_ok(T + 64, C, dil) == 0 and L.lib().ds_melgan_resblock_fused_ok(T, 128, dil) == 0
    x = rnd((B, T, C), "rb1.x%d.%d" % (dil, C), 2.0)
    w3, w2, ws = rnd((C, C, 3), "rb1.w3.%d" % C, 0.15), rnd((C, C), "rb1.w2.%d" % C, 0.2), rnd((C, C), "rb1.ws.%d" % C, 0.2)
    b3, b2, bs = rnd((C,), "rb1.b3.%d" % C), rnd((C,), "rb1.b2.%d" % C), rnd((C,), "rb1.bs.%d" % C)
    ref = _resblock_ref64(x, w3, b3, w2, b2, ws, bs, dil)
    p3, s3 = L.split_f16x2(w3.permute(0, 2, 1).reshape(C, 3 * C).contiguous().cuda())       # K ordered [tap][channel]
    pt, st = L.split_f16x2(torch.cat((w2, ws), 1).contiguous().cuda())
    xc, b3c, btc = x.cuda(), b3.cuda(), (b2 + bs).cuda()
    out = {}
    for name, hbuf in (("one", None), ("two", torch.empty(B, T, C, device="cuda"))):
        y = torch.full((B, T, C), float("nan"), device="cuda")
        L.check(L.lib().ds_melgan_resblock(L.ptr(xc), L.ptr(p3), C * 3 * C, s3, L.ptr(b3c), L.ptr(pt), C * 2 * C, st, L.ptr(btc),
                                           L.ptr(hbuf), L.ptr(y), B, T, C, dil, L.stream()))
        out[name] = y.cpu()
    e1, e2 = relerr(out["one"], ref), relerr(out["two"], ref)
    d = float((out["one"] - out["two"]).abs().max())
    print("MelGAN ResnetBlock C %d dil %d: single pass %.2e, two launches %.2e vs float64; max |one - two| %.2e" % (C, dil, e1, e2, d))
    assert torch.isfinite(out["one"]).all() and e1 < 3e-6 and e2 < 3e-6
    # a request the single-pass kernel is not built for fails loudly instead of falling back
    y = torch.empty(B, T + 64, C, device="cuda")
    xx = torch.zeros(B, T + 64, C, device="cuda")
    assert L.lib().ds_melgan_resblock(L.ptr(xx), L.ptr(p3), C * 3 * C, s3, L.ptr(b3c), L.ptr(pt), C * 2 * C, st, L.ptr(btc), None,
                                      L.ptr(y), B, T + 64, C, dil, L.stream()) != 0


def test_melgan_final_single_pass_and_generator_ab():
    """ds_melgan_final (LReLU + reflect k7 conv 32 -> 1 + tanh in one pass) against float64, ragged last tile and clip ends
    included; then the whole Generator with the single-pass kernels on and off."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd import _lib as L
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    B, T, C = 3, 1000, 32
    x = rnd((B, T, C), "fin.x", 2.0)
    w, bias = rnd((7, C), "fin.w", 0.1), 0.05
    xr = F.pad(F.leaky_relu(x.double().permute(0, 2, 1), 0.2), (3, 3), mode="reflect")
    ref = torch.tanh(F.conv1d(xr, w.double().t()[None], torch.tensor([bias], dtype=torch.float64)))[:, 0]
    xc, wc = x.cuda(), w.cuda()
    out = torch.full((B, T), float("nan"), device="cuda")
    L.check(L.lib().ds_melgan_final(L.ptr(xc), L.ptr(wc), bias, L.ptr(out), B, T, C, L.stream()))
    e = float((out.cpu().double() - ref).abs().max())
    print("MelGAN final layer, single pass: max abs error vs float64 %.2e" % e)
    assert torch.isfinite(out).all() and e < 2e-6
    assert L.lib().ds_melgan_final(L.ptr(xc), L.ptr(wc), bias, L.ptr(out), B, T, 64, L.stream()) != 0      # not built: loud


@pytest.mark.parametrize("C,T,dil", [(128, 700, 1), (128, 512, 9), (256, 300, 3), (128, 1000, 27)])
def test_conv1d_k3_halo_vs_float64_and_gather_kernel(C, T, dil):
    """ds_conv1d_k3_f16x2 (conv1d_f16x2.hip: MelGAN's dilated k3 conv with ReflectionPad1d, LeakyReLU in front) against float64
    and against the tap-by-tap kernel; ragged last tile, clip ends, one / two output-channel tiles, several slabs."""
    from text_to_sound_synthesis_amd import _lib as L
    B = 3
    x = rnd((B, T, C), "c1.x%d.%d" % (C, dil), 2.0)
    w, bias = rnd((C, C, 3), "c1.w%d" % C, 0.1), rnd((C,), "c1.b%d" % C)
    xr = F.pad(F.leaky_relu(x.double().permute(0, 2, 1), 0.2), (dil, dil), mode="reflect")
    ref = F.conv1d(xr, w.double(), bias.double(), dilation=dil).permute(0, 2, 1).contiguous()
    w2d = w.permute(0, 2, 1).reshape(C, 3 * C).contiguous().cuda()                      # K ordered [tap][channel]
    planes, sc = L.split_f16x2(w2d)
    wq = L.pack_conv_weights(planes, C, C, 3)
    xc, bc = x.cuda(), bias.cuda()
    y = torch.full((B, T, C), float("nan"), device="cuda")
    L.check(L.lib().ds_conv1d_k3_f16x2(L.ptr(xc), L.ptr(wq), wq.numel(), sc, L.ptr(bc), L.ptr(y), B, T, C, C, dil, 1, L.stream()))
    y2 = torch.empty(B, T, C, device="cuda")
    L.gemm(xc, planes, y2, B * T, C, 3 * C, split2=sc, conv_split=True, bias=bc, loader=L.LOAD_CONV1D, pro=L.PRO_LRELU, Cin=C, Wd=T,
           taps=3, dil=dil)
    e1, e2 = relerr(y.cpu(), ref), relerr(y2.cpu(), ref)
    print("conv1d k3 C %d T %d dil %d: halo kernel %.2e, gather kernel %.2e vs float64; max |a - b| %.2e"
          % (C, T, dil, e1, e2, float((y - y2).abs().max())))
    assert torch.isfinite(y).all() and e1 < 3e-6 and e1 <= 1.5 * e2 + 1e-7


@pytest.mark.parametrize("cin,cout,T", [(128, 64, 300), (64, 32, 1000)])
def test_melgan_convt2_single_pass(cin, cout, T):
    """ds_melgan_convt2: LeakyReLU + ConvTranspose1d(k = 4, s = 2, p = 1) in one pass against float64 (torch's conv_transpose1d)
    and against the polyphase GEMMs; ragged last tile, both clip ends."""
    from text_to_sound_synthesis_amd import _lib as L
    B = 3
    x = rnd((B, T, cin), "ct.x%d" % cin, 2.0)
    w, bias = rnd((cin, cout, 4), "ct.w%d" % cin, 0.1), rnd((cout,), "ct.b%d" % cin)
    ref = F.conv_transpose1d(F.leaky_relu(x.double().permute(0, 2, 1), 0.2), w.double(), bias.double(), stride=2, padding=1)
    ref = ref.permute(0, 2, 1).contiguous()                                      # [B][2 T][cout]
    # polyphase packing of modeling/vocoder.py: [r][Cout][2][Cin], phase p: taps p (on x[s0]) and p + r (on x[s0 - 1])
    wph = w.permute(2, 1, 0).reshape(2, 2, cout, cin).permute(1, 2, 0, 3).reshape(2, cout, 2 * cin).contiguous()
    planes, sc = L.split_f16x2(wph.reshape(-1, 2 * cin).cuda())
    xc, bc = x.cuda(), bias.cuda()
    y = torch.full((B, 2 * T, cout), float("nan"), device="cuda")
    assert L.lib().ds_melgan_convt2_ok(cin, cout) == 1 and L.lib().ds_melgan_convt2_ok(cin, 128) == 0
    L.check(L.lib().ds_melgan_convt2(L.ptr(xc), L.ptr(planes), 2 * cout * 2 * cin, sc, L.ptr(bc), L.ptr(y), B, T, cin, cout, L.stream()))
    y2 = torch.empty(B, 2 * T, cout, device="cuda")
    L.gemm(xc, planes, y2, B * T, cout, 2 * cin, split2=sc, conv_split=True, bias=bc, ldc=cout, loader=L.LOAD_CONVT1D, pro=L.PRO_LRELU,
           store=L.STORE_CONVT, groups=2, w_gstride=cout * 2 * cin, Cin=cin, Wd=T, ct_r=2, ct_p=1, ct_tin=T)
    e1, e2 = relerr(y.cpu(), ref), relerr(y2.cpu(), ref)
    print("ConvTranspose1d %d -> %d, T %d: single pass %.2e, polyphase GEMMs %.2e vs float64; max |a - b| %.2e"
          % (cin, cout, T, e1, e2, float((y - y2).abs().max())))
    assert torch.isfinite(y).all() and e1 < 3e-6
    assert L.lib().ds_melgan_convt2(L.ptr(xc), L.ptr(planes), 2 * cout * 2 * cin, sc, L.ptr(bc), L.ptr(y), B, T, cin, 128, L.stream()) != 0


def test_generator_default_kernels_vs_strict_fp32_mode():
    """The whole Generator on its default kernels (one-GEMM block tails, single-pass 32 / 64-channel blocks, halo-tiled k3
    and transposed convs, fused final layer) against the strict mode (every layer a gather-GEMM on the exact-fp32 MFMA)."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    g = Generator(80, 32, 3)
    g.load_state_dict(synth_sd("generator"))
    g = g.cuda().eval()
    mel = synth.synth_uniform((2, 80, 53), key="rt.mel").cuda()
    a = g(mel).cpu()
    g.conv_precision = "fp32"
    bb = g(mel).cpu()
    rms = float((a - bb).pow(2).mean().sqrt())
    print("MelGAN: default f16x2 kernels vs the exact-fp32 gather-GEMM forms, waveform RMS difference %.2e" % rms)
    assert torch.isfinite(a).all() and rms < 1e-6


@pytest.mark.parametrize("cin,cout,T,r", [(256, 128, 300, 8), (512, 256, 130, 8), (128, 128, 257, 2)])
def test_convt1d_halo_polyphase(cin, cout, T, r):
    """ds_convt1d_f16x2: LeakyReLU + ConvTranspose1d(k = 2 r, stride r, padding r / 2) as r two-tap convs over one staged input
    tile, against float64 (torch's conv_transpose1d) and against the polyphase GEMMs of the gather kernel."""
    from text_to_sound_synthesis_amd import _lib as L
    B, pad = 2, r // 2 + r % 2
    x = rnd((B, T, cin), "cth.x%d" % cin, 2.0)
    w, bias = rnd((cin, cout, 2 * r), "cth.w%d.%d" % (cin, r), 0.05), rnd((cout,), "cth.b%d" % cin)
    ref = F.conv_transpose1d(F.leaky_relu(x.double().permute(0, 2, 1), 0.2), w.double(), bias.double(), stride=r, padding=pad,
                             output_padding=r % 2).permute(0, 2, 1).contiguous()
    wph = w.permute(2, 1, 0).reshape(2, r, cout, cin).permute(1, 2, 0, 3).reshape(r, cout, 2 * cin).contiguous()
    planes, sc = L.split_f16x2(wph.reshape(-1, 2 * cin).cuda())
    pl = planes.view(2, r, cout, 2 * cin)
    wq = torch.cat([L.pack_conv_weights(pl[:, g].contiguous(), cout, cin, 2) for g in range(r)])
    xc, bc = x.cuda(), bias.cuda()
    y = torch.full((B, T * r, cout), float("nan"), device="cuda")
    L.check(L.lib().ds_convt1d_f16x2(L.ptr(xc), L.ptr(wq), wq.numel(), sc, L.ptr(bc), L.ptr(y), B, T, cin, cout, r, pad, 1, L.stream()))
    y2 = torch.empty(B, T * r, cout, device="cuda")
    L.gemm(xc, planes, y2, B * T, cout, 2 * cin, split2=sc, conv_split=True, bias=bc, ldc=cout, loader=L.LOAD_CONVT1D, pro=L.PRO_LRELU,
           store=L.STORE_CONVT, groups=r, w_gstride=cout * 2 * cin, Cin=cin, Wd=T, ct_r=r, ct_p=pad, ct_tin=T)
    e1, e2 = relerr(y.cpu(), ref), relerr(y2.cpu(), ref)
    print("ConvTranspose1d %d -> %d, r %d, T %d: halo kernel %.2e, gather kernel %.2e vs float64; max |a - b| %.2e"
          % (cin, cout, r, T, e1, e2, float((y - y2).abs().max())))
    assert torch.isfinite(y).all() and e1 < 3e-6
