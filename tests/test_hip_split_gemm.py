"""The fp32-class split GEMM (csrc/gemm_f16x2.hip: 2 fp16 planes, 3 MFMA passes) and the denoiser running on it: same
references, same tolerances as the fp32-MFMA path.  GPU only."""
import pytest
import torch

from conftest import golden, synth_sd
from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
NO_GRAD = True          # tests/conftest.py: every test of this module runs under torch.no_grad()


def rnd(shape, key, scale=1.0):
    return (synth.synth_uniform(shape, key=key) * 2 - 1) * scale


def relerr(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(530, 1024, 1024), (265, 3072, 1024), (300, 256, 1024), (530, 1024, 4096),
                                   (64, 96, 32), (1, 32, 64)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
@pytest.mark.parametrize("mode", ["f16x2"])
def test_split_gemm_matches_float64(M, N, K, tile, mode):
    from text_to_sound_synthesis_amd import _lib as L
    A, W, b, R = rnd((M, K), "sA", 3.0), rnd((N, K), "sW", 0.1), rnd((N,), "sb"), rnd((M, N), "sR")
    A[0, :8] = torch.tensor([1e-6, -3e4, 7.0, 1e-3, -1e-9, 0.0, 255.0, -0.5])   # wide dynamic range in one row
    W[0, :4] = torch.tensor([1e-7, -2e-5, 0.0, 3e-3])                           # tiny weights next to O(0.1) ones
    ref = (A.double() @ W.double().t() + b.double() + R.double()).float()
    out = torch.full((M, N), float("nan"), device="cuda")
    Ac, Wc, bc, Rc = A.cuda(), W.cuda(), b.cuda(), R.cuda()
    force = L.lib().ds_gemm_f16x2_force_tile
    force(tile)
    try:
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


@pytest.mark.parametrize("mode", ["f16x2"])
def test_split_gemm_gelu_and_transposed_store(mode):
    from text_to_sound_synthesis_amd import _lib as L
    B, Lr, N, K = 3, 265, 256, 1024
    M = B * Lr
    A, W, b = rnd((M, K), "sgA"), rnd((N, K), "sgW", 0.05), rnd((N,), "sgb")
    y = A.double() @ W.double().t() + b.double()
    Ac, bc = A.cuda(), b.cuda()
    W3, sc = L.split_f16x2(W.cuda())
    kw = dict(split2=sc)
    out = torch.empty(M, N, device="cuda")
    L.gemm(Ac, W3, out, M, N, K, bias=bc, act=L.ACT_GELU2, **kw)
    assert relerr(out.cpu(), (y * torch.sigmoid(1.702 * y)).float()) < 3e-6
    outT = torch.empty(B, N, Lr, device="cuda")
    L.gemm(Ac, W3, outT, M, N, K, bias=bc, ldc=Lr, store=L.STORE_BATCH_T, rows_per_sample=Lr, **kw)
    assert relerr(outT.cpu(), y.view(B, Lr, N).transpose(1, 2).float()) < 2e-6


def build(n_layer, T=100, mode="f16x2"):
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=n_layer, diffusion_step=T))
    sd = dict(synth_sd("dalle", n_layer))
    if T != 100:
        sd = {k: (v[:T] if k.endswith(("ln1.emb.weight", "ln1_1.emb.weight")) else v) for k, v in sd.items()}
    m.load_state_dict(sd, strict=False)
    m.transformer.transformer.precision = mode
    return m.cuda().eval()


@pytest.mark.parametrize("mode", ["f16x2"])
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


@pytest.mark.parametrize("mode", ["f16x2"])
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


def test_default_mode_step_is_batch_size_invariant():
    """Every batch size picks its own GEMM tile configs (64x64 / 128x64 / balanced 128x128 + 64x64 tail, per GEMM
    shape) and puts the sample boundaries at other places inside the tiles of the attention-ready stores: row 0 of
    a step over B identical samples must equal the B = 1 step bit for bit, and all rows must agree."""
    m = build(2, mode="f16x2")
    dt = m.transformer
    dt.truncation_r = 0.85
    x1 = synth.synth_tokens(1, mask_frac=0.5, key="bs.x").cuda()
    c1 = synth.synth_cond_emb(1, key="bs.c").cuda()
    u1 = synth.synth_uniform((1, 257, 265), key="bs.u").cuda()
    ref_tok = ref_logits = None
    for B in (1, 2, 3, 8, 17, 21, 33, 47, 64):
        x, c, u = x1.expand(B, -1).contiguous(), c1.expand(B, -1, -1).contiguous(), u1.expand(B, -1, -1).contiguous()
        t = torch.full((B,), 41, dtype=torch.long, device="cuda")
        logits = dt.transformer(x, c, t)
        kv = dt.transformer.condition_kv(c, dt._schedule_table())
        tok = dt.p_sample_tokens(x, kv, t, u, initial=False)
        if ref_tok is None:
            ref_tok, ref_logits = tok.clone(), logits.clone()
        assert torch.equal(logits, ref_logits.expand(B, -1, -1)), "logits differ at B=%d" % B
        assert torch.equal(tok, ref_tok.expand(B, -1)), "tokens differ at B=%d" % B


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_codebook_512_vs_reference(precision):
    """BASELINE configs[3] uses the 512-entry codebook (caps_512.yaml: 513 classes, logits N = 512): logits, one
    teacher-forced step and the decode of its tokens against the reference's outputs (tests/golden/k512_L2.npz)."""
    from text_to_sound_synthesis_amd.config import build_model, default_config
    g = golden("k512_L2")
    m = build_model(default_config(n_layer=2, diffusion_step=100, n_embed=512))
    sd = dict(synth_sd("dalle_k512", 2))
    sd.update(synth_sd("encoder"))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(".mask" in k or "shuffle_idx" in k or ".log_" in k or ".Lt_" in k for k in missing)
    m = m.cuda().eval()
    m.transformer.transformer.precision = precision
    dt = m.transformer
    dt.truncation_r = 0.85
    x = synth.synth_tokens(2, 265, 512, mask_frac=0.4, key="k512.x").cuda()
    cond = synth.synth_cond_emb(2, key="k512.c").cuda()
    t = torch.tensor([61, 12]).cuda()
    s = slice(None, None, int(g["pos_stride"]))
    out = dt.transformer(x, cond, t).cpu()
    assert out.shape == (2, 512, 265) and (out[:, :, s] - g["logits"]).abs().max() < 5e-5
    u = synth.synth_uniform((2, 513, 265), key="k512.u").cuda()
    tok, dump = dt.step_detail(x, cond, t, u, initial=False)
    assert (dump["log_pred"].cpu()[:, :, s] - g["log_pred"]).abs().max() < 1e-4
    assert ((dump["trunc"].cpu() > -70).sum(1) != g["kept"]).sum().item() == 0
    assert (dump["post"].cpu()[:, :, s] - g["post"]).abs().max() < 2e-4
    assert (tok.cpu() != g["tokens"]).sum().item() == 0
    mel = m.decode_to_img(g["tokens"][:1].clamp(max=511).cuda(), (1, 256, 5, 53)).cpu()
    assert (mel[0] - g["mel0"]).abs().max() < 1e-3


@pytest.mark.parametrize("Lk,B", [(265, 2), (77, 3), (32, 1), (288, 1)])
def test_attention_f16x2_matches_oracle(Lk, B):
    """ds_attention_f16x2 against the same reference and tolerance as the fp32-MFMA kernel."""
    import math
    import diffsound_oracle as O
    from text_to_sound_synthesis_amd import _lib as L
    Lq, H, D = 265, 16, 1024
    q, k, v = rnd((B, Lq, D), "at.q"), rnd((B, Lk, D), "at.k"), rnd((B, Lk, D), "at.v", 2.0)
    k[:, 5] *= 6.0                      # a spiked key: far-from-uniform softmax rows
    v[:, 3, :7] = torch.tensor([1e-6, -3e-5, 250.0, 0.0, -1e-3, 7.5, 1e-8])
    ref = O._mha(q.double(), k.double(), v.double(), H).float()
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    out = torch.full((B * Lq, D), float("nan"), device="cuda")
    L.check(L.lib().ds_attention_f16x2(L.ptr(qc), D, L.ptr(kc), D, L.ptr(vc), D, L.ptr(out), D, B, H, Lq, Lk,
                                       1.0 / math.sqrt(64), L.stream()))
    f32 = torch.empty(B * Lq, D, device="cuda")
    L.check(L.lib().ds_attention(L.ptr(qc), D, L.ptr(kc), D, L.ptr(vc), D, L.ptr(f32), D, B, H, Lq, Lk,
                                 1.0 / math.sqrt(64), L.stream()))
    e16, e32 = (out.cpu().view_as(ref) - ref).abs().max().item(), (f32.cpu().view_as(ref) - ref).abs().max().item()
    print("attention Lk=%d: f16x2 max-abs %.2e, fp32-MFMA max-abs %.2e (vs float64)" % (Lk, e16, e32))
    assert e16 < 2e-5


def test_attention_f16x2_strided_qkv_and_speed():
    import diffsound_oracle as O
    from text_to_sound_synthesis_amd import _lib as L
    B, Lq, H, D = 64, 265, 16, 1024
    qkv = torch.randn(B, Lq, 3 * D, device="cuda")
    out = torch.empty(B * Lq, D, device="cuda")
    args = (qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, 3 * D, qkv.data_ptr() + 8 * D, 3 * D, L.ptr(out), D, B, H,
            Lq, Lq, 0.125, L.stream())
    L.check(L.lib().ds_attention_f16x2(*args))
    ref = O._mha(qkv[:2, :, :D].cpu(), qkv[:2, :, D:2 * D].cpu(), qkv[:2, :, 2 * D:].cpu(), H)
    assert (out.view(B, Lq, D)[:2].cpu() - ref).abs().max() < 2e-5
    for name, fn in (("f16x2", L.lib().ds_attention_f16x2), ("fp32", L.lib().ds_attention)):
        for _ in range(3):
            fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        print("self-attention B=64 %s: %.1f us" % (name, e0.elapsed_time(e1) * 100))


# ---- packed split planes: producers write them, the GEMM stages them by LDS-DMA -------------------------------------
def torch_split(a):
    """ds_split_hi/lo (csrc/common.h) in torch: hi = fp16(clamp(a)), lo = fp16(clamp(a - hi))."""
    hi = a.clamp(-65504.0, 65504.0).half()
    lo = (a - hi.float()).clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, lo)).contiguous()


@pytest.mark.parametrize("M,N,K", [(530, 1024, 1024), (300, 256, 1024), (530, 1024, 4096), (64, 96, 32), (1, 32, 64),
                                   (2100, 1024, 1024)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3])
def test_f16x2_packed_operands_bit_identical(M, N, K, tile):
    """A and W handed over as packed split planes (LDS-DMA staging) give the same bits as the loader-split GEMM on
    row-major operands; C written packed is the packed split of C."""
    from text_to_sound_synthesis_amd import _lib as L
    A, W, b = rnd((M, K), "psA", 3.0), rnd((N, K), "psW", 0.1), rnd((N,), "psb")
    A[0, :6] = torch.tensor([1e-6, -3e4, 7e4, -1e5, -1e-9, 0.0])       # includes values past the fp16 range
    Ac, bc = A.cuda(), b.cuda()
    W2, sc = L.split_f16x2(W.cuda())
    W2p, scp = L.split_f16x2(W.cuda(), packed=True)
    assert sc == scp
    A2p = L.pack_planes(torch_split(Ac))
    M16 = (M + 15) // 16 * 16
    L.lib().ds_gemm_f16x2_force_tile(tile)
    try:
        for act in (L.ACT_NONE, L.ACT_GELU2):
            ref = torch.empty(M, N, device="cuda")
            L.gemm(Ac, W2, ref, M, N, K, bias=bc, act=act, split2=sc)
            out = torch.full((M, N), float("nan"), device="cuda")
            L.gemm(A2p, W2p, out, M, N, K, bias=bc, act=act, split2=sc, a_plane=M16 * K)
            assert torch.equal(out, ref)
            if N % 32 == 0:
                outs = torch.zeros(2, M16 * N, device="cuda", dtype=torch.float16)
                L.gemm(A2p, W2p, outs, M, N, K, bias=bc, act=act, split2=sc, a_plane=M16 * K, c_plane=M16 * N)
                assert torch.equal(L.unpack_planes(outs, M, N), torch_split(ref))
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)


@pytest.mark.parametrize("M,N,K,slots", [(530, 1024, 1024, 8), (2100, 1024, 1024, 16), (700, 256, 4096, 2), (1000, 96, 64, 1)])
def test_f16x2_balanced_hybrid_launch_bit_identical(M, N, K, slots):
    """128x128 tiles over the leading rows + 64x64 tiles over the tail rows == the loader-split GEMM, for row-major
    (with residual) and packed outputs; `slots` shrinks the balance unit so these small shapes take the hybrid path."""
    from text_to_sound_synthesis_amd import _lib as L
    A, W, b, R = rnd((M, K), "hyA", 3.0), rnd((N, K), "hyW", 0.1), rnd((N,), "hyb"), rnd((M, N), "hyR")
    Ac, bc, Rc = A.cuda(), b.cuda(), R.cuda()
    W2, sc = L.split_f16x2(W.cuda())
    W2p, _ = L.split_f16x2(W.cuda(), packed=True)
    A2p = L.pack_planes(torch_split(Ac))
    M16 = (M + 15) // 16 * 16
    ref = torch.empty(M, N, device="cuda")
    L.gemm(Ac, W2, ref, M, N, K, bias=bc, R=Rc, split2=sc)
    refg = torch.empty(M, N, device="cuda")
    L.gemm(Ac, W2, refg, M, N, K, bias=bc, act=L.ACT_GELU2, split2=sc)
    L.lib().ds_gemm_f16x2_force_tile(0)
    L.lib().ds_gemm_f16x2_set_balance_slots(slots)
    try:
        out = torch.full((M, N), float("nan"), device="cuda")
        L.gemm(A2p, W2p, out, M, N, K, bias=bc, R=Rc, split2=sc, a_plane=M16 * K)
        assert torch.equal(out, ref)
        if N % 32 == 0:
            outs = torch.zeros(2, M16 * N, device="cuda", dtype=torch.float16)
            L.gemm(A2p, W2p, outs, M, N, K, bias=bc, act=L.ACT_GELU2, split2=sc, a_plane=M16 * K, c_plane=M16 * N)
            assert torch.equal(L.unpack_planes(outs, M, N), torch_split(refg))
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)
        L.lib().ds_gemm_f16x2_set_balance_slots(512)


@pytest.mark.parametrize("M,N,K", [(4240, 4096, 1024), (4240, 2048, 1024), (4240, 1536, 1024)])
def test_f16x2_large_grids_all_launch_variants(M, N, K):
    """Grids of several rounds of resident workgroups (where DMA latency exceeds a k-tile's compute): every launch
    variant of the packed-operand GEMM -- 128x64, 64x64, plain 128x128, balanced hybrid with three balance units --
    must be bit-identical to the loader-split kernel.  (A dropped s_waitcnt vmcnt(0) made exactly these fail.)"""
    from text_to_sound_synthesis_amd import _lib as L
    A, W, b = rnd((M, K), "lg.A", 2.0).cuda(), rnd((N, K), "lg.W", 0.05).cuda(), rnd((N,), "lg.b").cuda()
    W2, sc = L.split_f16x2(W)
    W2p, _ = L.split_f16x2(W, packed=True)
    A2p = L.pack_planes(torch_split(A))
    M16 = (M + 15) // 16 * 16
    ref = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, ref, M, N, K, bias=b, split2=sc)
    assert relerr(ref.cpu(), (A.double() @ W.double().t() + b.double()).float().cpu()) < 2e-6
    try:
        for tile, slots in ((1, 512), (2, 512), (3, 512), (0, 1 << 30), (0, 512), (0, 64), (0, 1)):
            L.lib().ds_gemm_f16x2_force_tile(tile)
            L.lib().ds_gemm_f16x2_set_balance_slots(slots)
            for rep in range(2):
                out = torch.full((M, N), float("nan"), device="cuda")
                L.gemm(A2p, W2p, out, M, N, K, bias=b, split2=sc, a_plane=M16 * K)
                assert torch.equal(out, ref), "tile %d slots %d" % (tile, slots)
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)
        L.lib().ds_gemm_f16x2_set_balance_slots(512)


def test_split_producers_bit_identical():
    """ds_adaln_split / ds_layernorm_split / ds_attention_f16x2_split == packed split of the fp32-output kernels."""
    from text_to_sound_synthesis_amd import _lib as L
    M, Lr, D, H = 530, 265, 1024, 16
    M16 = (M + 15) // 16 * 16
    x = rnd((M, D), "pp.x", 4.0).cuda()
    tab = rnd((100, 2 * D), "pp.tab").cuda()
    t = torch.tensor([3, 97], dtype=torch.int64).cuda()
    g, b = rnd((D,), "pp.g").cuda(), rnd((D,), "pp.b").cuda()
    y, ys = torch.empty(M, D, device="cuda"), torch.zeros(2, M16 * D, device="cuda", dtype=torch.float16)
    L.check(L.lib().ds_adaln(L.ptr(x), L.ptr(y), M, Lr, D, L.ptr(tab), L.ptr(t), L.stream()))
    L.check(L.lib().ds_adaln_split(L.ptr(x), L.ptr(ys), M, Lr, D, L.ptr(tab), L.ptr(t), L.stream()))
    assert torch.equal(L.unpack_planes(ys, M, D), torch_split(y))
    ys.zero_()
    L.check(L.lib().ds_layernorm(L.ptr(x), L.ptr(y), M, D, L.ptr(g), L.ptr(b), L.stream()))
    L.check(L.lib().ds_layernorm_split(L.ptr(x), L.ptr(ys), M, D, L.ptr(g), L.ptr(b), L.stream()))
    assert torch.equal(L.unpack_planes(ys, M, D), torch_split(y))
    for Lk in (265, 77):
        q, k, v = rnd((2, Lr, D), "pp.q").cuda(), rnd((2, Lk, D), "pp.k").cuda(), rnd((2, Lk, D), "pp.v", 300.0).cuda()
        ys.zero_()
        args = (L.ptr(q), D, L.ptr(k), D, L.ptr(v), D)
        L.check(L.lib().ds_attention_f16x2(*args, L.ptr(y), D, 2, H, Lr, Lk, 0.125, L.stream()))
        L.check(L.lib().ds_attention_f16x2_split(*args, L.ptr(ys), D, 2, H, Lr, Lk, 0.125, L.stream()))
        assert torch.equal(L.unpack_planes(ys, M, D), torch_split(y))


@pytest.mark.parametrize("Lk,B", [(265, 2), (77, 3)])
def test_attention_ready_operands_bit_identical(Lk, B):
    """ds_attention_f16x2_ready on host-packed Q planes / K, V^T images, and on images made by ds_attn_pack_kv,
    equals ds_attention_f16x2_split on the fp32 tensors (same split values -> same bits)."""
    from text_to_sound_synthesis_amd import _lib as L
    Lq, H, D = 265, 16, 1024
    q, k, v = rnd((B, Lq, D), "ar.q").cuda(), rnd((B, Lk, D), "ar.k").cuda(), rnd((B, Lk, D), "ar.v", 50.0).cuda()
    M16 = (B * Lq + 15) // 16 * 16
    ref = torch.zeros(2, M16 * D, device="cuda", dtype=torch.float16)
    L.check(L.lib().ds_attention_f16x2_split(L.ptr(q), D, L.ptr(k), D, L.ptr(v), D, L.ptr(ref), D, B, H, Lq, Lk, 0.125,
                                             L.stream()))
    heads = lambda x, n: torch_split(x).view(2, B, n, H, 64).permute(0, 1, 3, 2, 4).contiguous()   # [2][B][H][n][64]
    qh = heads(q, Lq)
    nkey = L.lib().ds_attn_nkey(Lk)
    img = L.attn_images(heads(k, Lk), heads(v, Lk), nkey)
    out = torch.zeros_like(ref)
    L.check(L.lib().ds_attention_f16x2_ready(L.ptr(qh), B * H * Lq * 64, L.ptr(img), L.ptr(out), D, B, H, Lq, Lk, 0.125,
                                             L.stream()))
    assert torch.equal(out, ref)
    kv = torch.cat((k, v), dim=2).contiguous()                       # [B*Lk][2D] rows: K | V
    img2 = torch.full_like(img, float("nan"))
    L.check(L.lib().ds_attn_pack_kv(L.ptr(kv), 2 * D, D, L.ptr(img2), B, H, Lk, L.stream()))
    assert torch.equal(img2, img)


def test_gemm_attention_store_bit_identical():
    """The fused QKV projection written attention-ready (STORE_ATTN: Q planes + K / V^T images, staged through LDS and
    stored in 16-byte pieces) equals the same GEMM written row-major, split and packed on the host; also through
    the balanced hybrid launch (absolute-row bookkeeping, 64x64 tail tiles) and for sample boundaries inside tiles."""
    from text_to_sound_synthesis_amd import _lib as L
    B, Lq, H, D = 3, 265, 16, 1024
    M, N, K = B * Lq, 3 * D, D
    A, W, b = rnd((M, K), "as.A", 2.0).cuda(), rnd((N, K), "as.W", 0.05).cuda(), rnd((N,), "as.b").cuda()
    W2p, sc = L.split_f16x2(W, packed=True)
    A2p = L.pack_planes(torch_split(A))
    M16 = (M + 15) // 16 * 16
    ref = torch.empty(M, N, device="cuda")
    L.gemm(A2p, W2p, ref, M, N, K, bias=b, split2=sc, a_plane=M16 * K)
    heads = lambda x: torch_split(x.contiguous()).view(2, B, Lq, H, 64).permute(0, 1, 3, 2, 4).contiguous()
    q_ref = heads(ref[:, :D])
    img_ref = L.attn_images(heads(ref[:, D:2 * D]), heads(ref[:, 2 * D:]), 288)
    Wq2p, scq = L.split_f16x2(W[:D].contiguous(), packed=True)        # the cross-attention query projection alone
    refq = torch.empty(M, D, device="cuda")
    L.gemm(A2p, Wq2p, refq, M, D, K, bias=b, split2=scq, a_plane=M16 * K)
    for slots, tile in ((512, -1), (8, 0), (512, 1), (512, 2), (512, 3)):     # (8, 0): these shapes take the hybrid path
        L.lib().ds_gemm_f16x2_set_balance_slots(slots)
        L.lib().ds_gemm_f16x2_force_tile(tile)
        try:
            qh = torch.full((2, B, H, Lq, 64), float("nan"), device="cuda", dtype=torch.float16)
            img = torch.zeros(B, H, 4, 288 * 64, device="cuda", dtype=torch.float16)
            L.gemm(A2p, W2p, qh, M, N, K, bias=b, split2=sc, a_plane=M16 * K, store=L.STORE_ATTN, rows_per_sample=Lq,
                   attn=(img, H, 288, B * H * Lq * 64))
            assert torch.equal(qh, q_ref)
            assert torch.equal(img[:, :, :2], img_ref[:, :, :2])          # K image
            assert torch.equal(img[:, :, 2:], img_ref[:, :, 2:])          # V^T image (padding keys stay zero)
            # Q alone (the cross-attention query projection): N = heads * 64, no images
            qh.fill_(float("nan"))
            L.gemm(A2p, Wq2p, qh, M, D, K, bias=b, split2=scq, a_plane=M16 * K, store=L.STORE_ATTN, rows_per_sample=Lq,
                   attn=(None, H, 288, B * H * Lq * 64))
            assert torch.equal(qh, heads(refq))
        finally:
            L.lib().ds_gemm_f16x2_set_balance_slots(512)
            L.lib().ds_gemm_f16x2_force_tile(-1)


def test_packed_gemm_argument_checks():
    from text_to_sound_synthesis_amd import _lib as L
    A2 = torch.zeros(2, 16 * 32, device="cuda", dtype=torch.float16)
    W2, sc = L.split_f16x2(torch.ones(32, 32, device="cuda"), packed=True)
    out = torch.empty(8, 32, device="cuda")
    with pytest.raises(L.DiffsoundHipError):        # plane stride smaller than ceil16(M) * K
        L.gemm(A2, W2, out, 8, 32, 32, split2=sc, a_plane=8 * 32)
    with pytest.raises(L.DiffsoundHipError):        # a packed output cannot take a residual
        L.gemm(A2, W2, out, 8, 32, 32, R=out, split2=sc, a_plane=16 * 32, c_plane=16 * 32)
    L.gemm(A2, W2, out, 8, 32, 32, split2=sc, a_plane=16 * 32)
    assert torch.equal(out, torch.zeros_like(out))


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 1, 1, 1), (2, 3, 31, 16), (1, 2, 33, 17), (2, 16, 64, 96), (1, 4, 100, 97), (2, 2, 265, 128),
                                       (1, 3, 272, 191), (1, 16, 265, 192), (2, 5, 40, 193), (1, 2, 272, 288), (3, 1, 7, 250)])
def test_streamed_attention_shape_sweep(B, H, Lq, Lk):
    """The streamed (chunked, online-softmax) attention kernel on ragged shapes: query counts that leave waves of the last
    workgroup idle, key counts on and around the 96-key chunk boundaries (one / two / three chunks, last chunk full, one key
    into a chunk, one short of it), head counts that are not 16 -- the generic entry (fp32 operands, converted in the kernel),
    its packed-plane output form and the attention-ready entry (Q planes, K / V^T images by LDS-DMA), all against float64 and
    against one another (bit-identical: same arithmetic, different staging)."""
    import diffsound_oracle as O
    from text_to_sound_synthesis_amd import _lib as L
    D = H * 64
    q, k, v = rnd((B, Lq, D), "sw.q"), rnd((B, Lk, D), "sw.k"), rnd((B, Lk, D), "sw.v", 2.0)
    k[:, Lk // 2] *= 5.0                                     # one dominant key: far-from-uniform rows
    ref = O._mha(q.double(), k.double(), v.double(), H).float()
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    out = torch.full((B * Lq, D), float("nan"), device="cuda")
    L.check(L.lib().ds_attention_f16x2(L.ptr(qc), D, L.ptr(kc), D, L.ptr(vc), D, L.ptr(out), D, B, H, Lq, Lk, 0.125, L.stream()))
    err = (out.cpu().view_as(ref) - ref).abs().max().item()
    assert torch.isfinite(out).all() and err < 2e-5, err
    if D % 32 == 0:
        M16 = (B * Lq + 15) // 16 * 16
        sp = torch.zeros(2, M16 * D, device="cuda", dtype=torch.float16)
        L.check(L.lib().ds_attention_f16x2_split(L.ptr(qc), D, L.ptr(kc), D, L.ptr(vc), D, L.ptr(sp), D, B, H, Lq, Lk, 0.125, L.stream()))
        assert torch.equal(L.unpack_planes(sp, B * Lq, D), torch_split(out))
        heads = lambda x, n: torch_split(x).view(2, B, n, H, 64).permute(0, 1, 3, 2, 4).contiguous()
        qh = heads(qc, Lq)
        img = L.attn_images(heads(kc, Lk), heads(vc, Lk), L.lib().ds_attn_nkey(Lk))
        rd = torch.zeros_like(sp)
        L.check(L.lib().ds_attention_f16x2_ready(L.ptr(qh), B * H * Lq * 64, L.ptr(img), L.ptr(rd), D, B, H, Lq, Lk, 0.125, L.stream()))
        assert torch.equal(rd, sp)
