"""The fp32-class split GEMMs (csrc/gemm_bf16x3.hip: 3 bf16 planes, 6 MFMA passes; csrc/gemm_f16x2.hip: 2 fp16
planes, 3 passes) and the denoiser running on them: same references, same tolerances as the fp32-MFMA path.
GPU only."""
import pytest
import torch

from conftest import golden, synth_sd
from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(shape, key, scale=1.0):
    return (synth.synth_uniform(shape, key=key) * 2 - 1) * scale


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(530, 1024, 1024), (265, 3072, 1024), (300, 256, 1024), (530, 1024, 4096),
                                   (64, 96, 32), (1, 32, 64)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_split_gemm_matches_float64(M, N, K, tile, mode):
    from text_to_sound_synthesis_amd import _lib as L
    A, W, b, R = rnd((M, K), "sA", 3.0), rnd((N, K), "sW", 0.1), rnd((N,), "sb"), rnd((M, N), "sR")
    A[0, :8] = torch.tensor([1e-6, -3e4, 7.0, 1e-3, -1e-9, 0.0, 255.0, -0.5])   # wide dynamic range in one row
    W[0, :4] = torch.tensor([1e-7, -2e-5, 0.0, 3e-3])                           # tiny weights next to O(0.1) ones
    ref = (A.double() @ W.double().t() + b.double() + R.double()).float()
    out = torch.full((M, N), float("nan"), device="cuda")
    Ac, Wc, bc, Rc = A.cuda(), W.cuda(), b.cuda(), R.cuda()
    force = L.lib().ds_gemm_bf16x3_force_tile if mode == "bf16x3" else L.lib().ds_gemm_f16x2_force_tile
    force(tile)
    try:
        if mode == "bf16x3":
            W3 = L.split_bf16x3(Wc)
            assert torch.equal(W3.view(torch.bfloat16).float().sum(0), Wc)        # the split is exact
            L.gemm(Ac, W3, out, M, N, K, bias=bc, R=Rc, split3=True)
        else:
            W2, sc = L.split_f16x2(Wc)
            rec = W2.view(torch.float16).double().sum(0) * sc
            assert ((rec - Wc.double()).abs() <= 2.0 ** -22 * Wc.double().abs() + 2.0 ** -25 * sc).all()
            L.gemm(Ac, W2, out, M, N, K, bias=bc, R=Rc, split2=sc)
    finally:
        force(-1)
    f32 = torch.empty(M, N, device="cuda")
    L.gemm(Ac, Wc, f32, M, N, K, bias=bc, R=Rc)
    e3, e1 = relerr(out.cpu(), ref), relerr(f32.cpu(), ref)
    print("M%d N%d K%d tile %d: %s rel err %.2e, fp32-MFMA rel err %.2e" % (M, N, K, tile, mode, e3, e1))
    assert e3 < max(2e-6, 1.2 * e1)           # at least as accurate as the exact-fp32 FMA chain


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_split_gemm_gelu_and_transposed_store(mode):
    from text_to_sound_synthesis_amd import _lib as L
    B, Lr, N, K = 3, 265, 256, 1024
    M = B * Lr
    A, W, b = rnd((M, K), "sgA"), rnd((N, K), "sgW", 0.05), rnd((N,), "sgb")
    y = A.double() @ W.double().t() + b.double()
    Ac, bc = A.cuda(), b.cuda()
    if mode == "bf16x3":
        W3, kw = L.split_bf16x3(W.cuda()), dict(split3=True)
    else:
        W3, sc = L.split_f16x2(W.cuda())
        kw = dict(split2=sc)
    out = torch.empty(M, N, device="cuda")
    L.gemm(Ac, W3, out, M, N, K, bias=bc, act=L.ACT_GELU2, **kw)
    assert relerr(out.cpu(), (y * torch.sigmoid(1.702 * y)).float()) < 3e-6
    outT = torch.empty(B, N, Lr, device="cuda")
    L.gemm(Ac, W3, outT, M, N, K, bias=bc, ldc=Lr, store=L.STORE_BATCH_T, rows_per_sample=Lr, **kw)
    assert relerr(outT.cpu(), y.view(B, Lr, N).transpose(1, 2).float()) < 2e-6


def build(n_layer, T=100, mode="bf16x3"):
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=n_layer, diffusion_step=T))
    sd = dict(synth_sd("dalle", n_layer))
    if T != 100:
        sd = {k: (v[:T] if k.endswith(("ln1.emb.weight", "ln1_1.emb.weight")) else v) for k, v in sd.items()}
    m.load_state_dict(sd, strict=False)
    m.transformer.transformer.precision = mode
    return m.cuda().eval()


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_denoiser_split_vs_reference_logits(mode):
    m = build(2, mode=mode)
    tok = synth.synth_tokens(2, mask_frac=0.3, key="tf2.tokens").cuda()
    cond = synth.synth_cond_emb(2, key="tf2.cond").cuda()
    out = m.transformer.transformer(tok, cond, torch.tensor([37, 80]).cuda()).cpu()
    e = (out - golden("transformer_L2")["logits"]).abs().max().item()
    print("%s 2-layer logits max-abs vs reference: %.2e" % (mode, e))
    assert e < 5e-5
    del m
    m = build(19, mode=mode)
    tok = synth.synth_tokens(1, mask_frac=0.5, key="tf19.tokens").cuda()
    cond = synth.synth_cond_emb(1, key="tf19.cond").cuda()
    out = m.transformer.transformer(tok, cond, torch.tensor([63]).cuda()).cpu()
    e = (out - golden("transformer_L19")["logits"]).abs().max().item()
    print("%s 19-layer logits max-abs vs reference: %.2e" % (mode, e))
    assert e < 3e-4


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_denoiser_split_steps_and_trajectory_tokens_exact(mode):
    g = golden("steps_L2")
    m = build(2, mode=mode)
    dt = m.transformer
    dt.truncation_r = 0.85
    cond = synth.synth_cond_emb(1, key="step.cond").cuda()
    for tt, mf in ((99, None), (50, 0.55), (1, 0.02), (0, 0.0)):
        xt = torch.full((1, 265), 256) if mf is None else synth.synth_tokens(1, mask_frac=mf, key="step%d.xt" % tt)
        u = synth.synth_uniform((1, 257, 265), key="step%d.u" % tt)
        tok, d = dt.step_detail(xt.cuda(), cond, torch.tensor([tt]).cuda(), u.cuda(), initial=mf is None)
        assert (d["log_pred"].cpu()[:, :, ::int(g["pos_stride"])] - g["t%d_log_pred" % tt]).abs().max() < 1e-4
        assert (tok.cpu() != g["t%d_tokens" % tt]).sum().item() == 0
    del m
    g = golden("traj_T10_L2")
    m = build(2, T=10, mode=mode)
    m.transformer.truncation_r = 0.85
    cond = synth.synth_cond_emb(2, key="traj.cond").cuda()
    out = m.transformer.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0,
                               noise_fn=lambda t, shp: synth.synth_uniform(shp, key="traj.u%d" % t))
    assert (out["content_token"].cpu() != g["tokens"]).sum().item() == 0
