"""Module-level parity of the HIP path: drop-in modules (reference signatures + state-dict keys)
against the committed golden vectors (produced by the unmodified reference) and the oracle.
GPU only (-m gpu)."""
import pytest
import torch

import diffsound_oracle as O
from conftest import golden, synth_sd
from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
NO_GRAD = True          # tests/conftest.py: every test of this module runs under torch.no_grad()

MEL_TOL = 1e-3      # BASELINE.json north_star: max-abs on mel
WAVE_RMS_TOL = 1e-4  # BASELINE.json north_star: RMS on waveform


def build(n_layer, T=100):
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=n_layer, diffusion_step=T))
    sd = dict(synth_sd("dalle", n_layer))
    sd.update(synth_sd("encoder"))
    if T != 100:   # the timestep-embedding tables have one row per step (values are a prefix)
        sd = {k: (v[:T] if k.endswith(("ln1.emb.weight", "ln1_1.emb.weight")) else v) for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(".mask" in k or "shuffle_idx" in k or ".log_" in k or ".Lt_" in k for k in missing)
    m.transformer.transformer.precision = "fp32"   # this file pins the exact-fp32 MFMA path; the split
    return m.cuda().eval()                         # GEMM modes are covered by test_hip_split_gemm.py


@pytest.fixture(scope="module")
def m2():
    return build(2)


@pytest.fixture(scope="module")
def voc():
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    g = Generator(80, 32, 3)
    g.load_state_dict(synth_sd("generator"))
    return g.cuda().eval()


def test_schedule_buffers_match_reference(m2):
    g = golden("schedule")
    for n in ("log_at", "log_bt", "log_ct", "log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct", "log_1_min_ct",
              "log_1_min_cumprod_ct"):
        a, b = getattr(m2.transformer, n).cpu(), g["T100_" + n]
        fin = ~torch.isinf(b)
        assert torch.equal(torch.isinf(a), torch.isinf(b)) and torch.equal(a[fin], b[fin])


def test_transformer_L2_vs_reference(m2):
    tok = synth.synth_tokens(2, mask_frac=0.3, key="tf2.tokens").cuda()
    cond = synth.synth_cond_emb(2, key="tf2.cond").cuda()
    out = m2.transformer.transformer(tok, cond, torch.tensor([37, 80]).cuda()).cpu()
    ref = golden("transformer_L2")["logits"]
    assert out.shape == ref.shape
    assert (out - ref).abs().max() < 5e-5


def test_transformer_L19_vs_reference():
    m = build(19)
    tok = synth.synth_tokens(1, mask_frac=0.5, key="tf19.tokens").cuda()
    cond = synth.synth_cond_emb(1, key="tf19.cond").cuda()
    out = m.transformer.transformer(tok, cond, torch.tensor([63]).cuda()).cpu()
    ref = golden("transformer_L19")["logits"]
    assert (out - ref).abs().max() < 3e-4
    del m
    torch.cuda.empty_cache()


def test_teacher_forced_steps_vs_reference(m2):
    g = golden("steps_L2")
    ps = int(g["pos_stride"])
    dt = m2.transformer
    dt.truncation_r = 0.85
    cond = synth.synth_cond_emb(1, key="step.cond").cuda()
    for tt, mf in ((99, None), (50, 0.55), (1, 0.02), (0, 0.0)):
        xt = torch.full((1, 265), 256) if mf is None else synth.synth_tokens(1, mask_frac=mf, key="step%d.xt" % tt)
        u = synth.synth_uniform((1, 257, 265), key="step%d.u" % tt)
        tok, d = dt.step_detail(xt.cuda(), cond, torch.tensor([tt]).cuda(), u.cuda(), initial=mf is None)
        s = slice(None, None, ps)
        e_lp = (d["log_pred"].cpu()[:, :, s] - g["t%d_log_pred" % tt]).abs().max().item()
        kept = (d["trunc"].cpu() > -70).sum(1)
        n_kept = (kept != g["t%d_kept" % tt]).sum().item()
        e_tr = (d["trunc"].cpu()[:, :, s] - g["t%d_trunc" % tt]).abs().max().item()
        e_po = (d["post"].cpu()[:, :, s] - g["t%d_post" % tt]).abs().max().item()
        n_tok = (tok.cpu() != g["t%d_tokens" % tt]).sum().item()
        msg = "t=%d: log_pred %.2e, kept-count mismatches %d, trunc %.2e, post %.2e, token mismatches %d" % (
            tt, e_lp, n_kept, e_tr, e_po, n_tok)
        print(msg)
        # a class whose preceding mass is within float rounding of r may be kept/dropped differently
        assert e_lp < 1e-4 and n_kept <= 1 and n_tok <= 1, msg
        if n_kept == 0:
            assert e_tr < 1e-4 and e_po < 2e-4, msg


def test_trajectory_T10_then_decode_vocode_vs_reference(voc):
    """BASELINE config 1 without CLIP: 10 steps -> tokens -> mel -> waveform, noise injected."""
    g = golden("traj_T10_L2")
    m = build(2, T=10)
    m.transformer.truncation_r = 0.85
    cond = synth.synth_cond_emb(2, key="traj.cond").cuda()
    out = m.transformer.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0,
                               noise_fn=lambda t, shp: synth.synth_uniform(shp, key="traj.u%d" % t))
    tokens = out["content_token"]
    assert (tokens.cpu() != g["tokens"]).sum().item() == 0
    mel = m.decode_to_img(tokens, (2, 256, 5, 53))
    assert mel.shape == (2, 1, 80, 848)
    assert (mel[0].cpu() - g["mel0"]).abs().max() < MEL_TOL
    wave = voc(mel[:, 0], scale=0.5, shift=0.5)
    assert wave.shape == (2, 1, 217088)
    assert (wave[0, 0, :65536].cpu() - g["wave0_head"]).pow(2).mean().sqrt() < WAVE_RMS_TOL


def test_alternative_samplers_vs_reference():
    """SURVEY.md 8f-4 through generate_content's sample_type mini-language: 'top100p' (top-k), 'top0.85r,fast2'
    (skip-step), 'top0.85r,q0.5' (repeat-step) reproduce the reference's tokens with the same injected noise."""
    import random
    g = golden("samplers_T10_L2")
    cond = synth.synth_cond_emb(2, key="traj.cond").cuda()

    def run(sample_type, noise_fn, seed=None):
        m = build(2, T=10)                      # fresh model: the truncation choice is sticky, as in the reference
        tr = m.transformer
        if seed is not None:
            random.seed(seed)
        orig = {"sample": tr.sample, "sample_fast": tr.sample_fast}
        tr.sample = lambda **kw: orig["sample"](noise_fn=noise_fn, **kw)
        tr.sample_fast = lambda **kw: orig["sample_fast"](noise_fn=noise_fn, **kw)
        out = m.generate_content(batch={"condition_embed_token": cond}, filter_ratio=0, replicate=1, content_ratio=1,
                                 sample_type=sample_type)
        assert out["content"].shape == (2, 1, 80, 848)
        return m, out["content_token"].cpu()

    m, tok = run("top100p", lambda t, shp: synth.synth_uniform(shp, key="topk.u%d" % (9 - t)))
    assert (tok != g["topk_tokens"]).sum().item() == 0
    assert m.transformer.truncation_k == 100 and m.transformer.truncation_r is None
    # the wrapped predict_start (top-k) on a half-masked state
    log_z = torch.log(torch.nn.functional.one_hot(synth.synth_tokens(2, mask_frac=0.5, key="topk.xt"), 257)
                      .permute(0, 2, 1).float().clamp(min=1e-30)).cuda()
    trunc = m.transformer.predict_start(log_z, cond, torch.tensor([5, 5]).cuda()).cpu()
    ref, s = g["topk_trunc"], slice(None, None, int(g["pos_stride"]))
    assert ((trunc[:, :, s] > -70) == (ref > -70)).all() and (trunc[:, :, s] - ref).abs().max() < 2e-4
    order = {9: 0, 6: 1, 3: 2, 0: 3}
    _, tok = run("top0.85r,fast2", lambda t, shp: synth.synth_uniform(shp, key="fast.u%d" % order[t]))
    assert (tok != g["fast2_tokens"]).sum().item() == 0
    calls = []
    _, tok = run("top0.85r,q0.5", lambda c, shp: (calls.append(c), synth.synth_uniform(shp, key="rep.u%d" % c))[1], seed=7)
    assert len(calls) == int(g["q05_calls"])
    assert (tok != g["q05_tokens"]).sum().item() == 0


def test_vq_encode_get_tokens_and_partial_resample_vs_reference():
    """VQModel.encode / DALLE.get_tokens / prepare_content on the HIP path, and sample() with
    filter_ratio > 0 (q_sample + reverse chain) -- all against the reference's outputs."""
    g = golden("encoder_T10_L2")
    m = build(2, T=10)
    m.transformer.truncation_r = 0.85
    mel = (synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1).cuda()
    h = m.content_codec.encode_latent(mel).cpu()
    err = (h - g["h"]).abs().max().item()
    print("encoder latent max-abs err vs reference %.2e (|h| max %.2f)" % (err, g["h"].abs().max()))
    assert h.shape == (2, 256, 5, 53) and err < 1e-4
    quant, loss, info = m.content_codec.encode(mel)
    idx = info[2].view(2, -1).cpu()
    clear = g["gap"] > 1e-3                       # the best two codes are further apart than the latent's error allows
    assert clear.float().mean() > 0.7 and torch.equal(idx[clear], g["indices"][clear])
    assert quant.shape == (2, 256, 5, 53) and info[1].shape == (530, 256) and torch.isfinite(loss)
    E = m.content_codec.quantize.embedding.weight
    assert torch.equal(quant.permute(0, 2, 3, 1).reshape(-1, 256), E[info[2][:, 0]])       # z_q rows are code vectors
    qz, tokens = m.get_tokens(mel)
    cm = clear.view(2, 5, 53).transpose(1, 2).reshape(2, -1)
    assert torch.equal(tokens.cpu()[cm], g["tokens"][cm])
    pc = m.prepare_content({"image": mel})
    assert torch.equal(pc["content_token"], tokens) and pc["content_quant"].shape == (2, 256, 5, 53)
    # the nearest-code search itself, on the reference's latent: exact except at rounding-level ties
    _, _, info2 = m.content_codec.quantize(g["h"].cuda())
    tight = g["gap"] > 1e-5
    assert torch.equal(info2[2].view(2, -1).cpu()[tight], g["indices"][tight])
    # partial re-sampling from the reference's tokens (filter_ratio 0.5 of T = 10)
    cond = synth.synth_cond_emb(2, key="traj.cond").cuda()
    calls = []
    out = m.transformer.sample(condition_token=None, condition_mask=None, condition_embed=cond,
                               content_token=g["tokens"].cuda(), filter_ratio=0.5,
                               noise_fn=lambda c, shp: (calls.append(c), synth.synth_uniform(shp, key="part.u%d" % c))[1])
    assert calls == list(range(int(g["partial_calls"])))
    assert (out["content_token"].cpu() != g["partial_tokens"]).sum().item() == 0


def test_training_loss_forward_vs_reference():
    """SURVEY.md 8f-3, forward value only: DiffusionTransformer.forward(return_loss=True) with the reference's
    timesteps and q_sample noise injected reproduces its loss, modelled posterior and Lt_history update; through
    DALLE.forward the content comes from the VQ encoder (prepare_input)."""
    g = golden("train_loss_L2")
    m = build(2, T=100)
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]   # caps_text.yaml
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0").cuda()
    cond = synth.synth_cond_emb(3, key="tl.c").cuda()
    t = torch.tensor([57, 0, 93]).cuda()
    dt.sample_time = lambda b, device, method="uniform": (t, torch.ones(3, device="cuda") / 100)
    u = synth.synth_uniform((3, 257, 265), key="tl.u")
    out = dt({"content_token": x0, "condition_embed_token": cond}, return_loss=True, noise=u)
    print("train loss %.6f vs reference %.6f" % (out["loss"].item(), float(g["loss"])))
    assert abs(out["loss"].item() - float(g["loss"])) < 2e-4 * float(g["loss"])
    s = slice(None, None, int(g["pos_stride"]))
    assert (out["logits"].cpu()[:, :, s] - g["model_prob"]).abs().max() < 2e-5
    assert torch.allclose(dt.Lt_history.cpu(), g["Lt_history"], rtol=5e-4, atol=1e-6)
    assert torch.equal(dt.Lt_count.cpu(), g["Lt_count"])
    # the batch-level entry: mel -> VQ tokens -> loss (finite, right shapes)
    del dt.sample_time
    mel = (synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1).cuda()
    out = m({"image": mel, "condition_embed_token": synth.synth_cond_emb(2, key="tl.c2").cuda()}, return_loss=True)
    assert out["logits"].shape == (2, 257, 265) and torch.isfinite(out["loss"]) and out["loss"].item() > 0


def test_q_sample_matches_oracle():
    import diffsound_oracle as O
    m = build(2, T=100)
    dt = m.transformer
    x0 = synth.synth_tokens(3, mask_frac=0.2, key="qs.x0")
    for tt in (0, 37, 99):
        t = torch.full((3,), tt, dtype=torch.long)
        u = synth.synth_uniform((3, 257, 265), key="qs.u%d" % tt)
        ref = O.q_sample(O.make_schedule(100, 257), x0, t, u, 257).argmax(1)
        got = dt.q_sample_tokens(x0.cuda(), t.cuda(), u.cuda()).cpu()
        assert (got != ref).sum().item() == 0


def test_sample_tail_top_k_argument_checks():
    from text_to_sound_synthesis_amd import _lib as L
    z = torch.zeros(265, 256, device="cuda")
    x = torch.zeros(1, 265, dtype=torch.long, device="cuda")
    t = torch.zeros(1, dtype=torch.long, device="cuda")
    u = torch.rand(1, 257, 265, device="cuda")
    sched = torch.zeros(8, 101, device="cuda")
    with pytest.raises(L.DiffsoundHipError):     # top-k and top-r are exclusive
        L.check(L.lib().ds_sample_tail_ex(L.ptr(z), L.ptr(x), L.ptr(t), L.ptr(u), L.ptr(sched), L.ptr(x), None, None,
                                          None, 1, 265, 256, 100, 0, 0.85, 10, L.stream()))


def test_decode_vs_reference_and_api_forms(m2):
    tok = synth.synth_tokens(1, mask_frac=0.0, key="dec.tokens").cuda()
    ref = golden("decode")["mel"]
    mel = m2.decode_to_img(tok, (1, 256, 5, 53)).cpu()
    assert (mel - ref).abs().max() < MEL_TOL
    # the reference's two-call form: permuter -> get_codebook_entry -> VQModel.decode
    idx = m2.first_stage_permuter(tok, reverse=True)
    q = m2.content_codec.quantize.get_codebook_entry(idx.reshape(-1), shape=(1, 5, 53, 256))
    assert q.shape == (1, 256, 5, 53)
    assert (m2.content_codec.decode(q).cpu() - ref).abs().max() < MEL_TOL


def test_decoder_stages_vs_oracle(m2):
    """Localises a decoder mismatch: mid / per-level activations against the oracle's taps."""
    sd = synth_sd("dalle", 2)
    tok = synth.synth_tokens(2, mask_frac=0.0, key="dec2.tokens")
    taps = {}
    ref = O.vq_decode(sd, O.codebook_gather(sd, tok), taps=taps)
    got = m2.decode_to_img(tok.cuda(), (2, 256, 5, 53)).cpu()
    assert (got - ref).abs().max() < MEL_TOL


def test_decode_full_batch_chunk_is_sample_independent(m2):
    """BASELINE configs[2] decodes 64 clips in one chunk (4.3 M implicit-GEMM rows at full resolution): every sample
    of a batch of identical tokens must equal the single-sample decode bit for bit (no cross-sample leakage, no
    32-bit index overflow)."""
    tok1 = synth.synth_tokens(1, mask_frac=0.0, key="dec.tokens").cuda()
    one = m2.decode_to_img(tok1, (1, 256, 5, 53))
    assert m2.content_codec.decode_chunk >= 64
    many = m2.decode_to_img(tok1.expand(64, -1).contiguous(), (64, 256, 5, 53))
    assert many.shape == (64, 1, 80, 848)
    assert torch.equal(many, one.expand(64, -1, -1, -1))
    del many
    torch.cuda.empty_cache()


def test_vocoder_vs_reference(voc):
    mel01 = synth.synth_uniform((1, 80, 848), key="voc.mel").cuda()
    wave = voc(mel01).cpu()
    ref = golden("vocoder")["wave"]
    assert wave.shape == ref.shape
    assert (wave - ref).pow(2).mean().sqrt() < WAVE_RMS_TOL
    assert (wave - ref).abs().max() < 1e-3


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_vocoder_short_and_batched(voc, sd_vocoder, precision):
    """Both arithmetic modes of the MelGAN stack (every Conv1d / ConvTranspose1d / 1x1 conv on the 3-pass fp16 split of
    conv_f16x2.hip, or on the exact-fp32 MFMA) against the CPU oracle: other length, batch > 1 (reflection padding at
    every sample edge, polyphase phases, LeakyReLU prologue, residual epilogue)."""
    mel = synth.synth_uniform((3, 80, 53), key="voc.short")
    ref = O.melgan_generator(sd_vocoder, mel)
    old = voc.conv_precision
    try:
        voc.conv_precision = precision
        got = voc(mel.cuda()).cpu()
    finally:
        voc.conv_precision = old
    assert got.shape == ref.shape == (3, 1, 53 * 256)
    rms = (got - ref).pow(2).mean().sqrt().item()
    print("vocoder %s: wave RMS vs oracle %.2e" % (precision, rms))
    assert rms < WAVE_RMS_TOL


def test_full_size_properties_B32():
    """BASELINE configs[1] size (B=32, K=256, 19 layers): properties that do not need the oracle."""
    m = build(19)
    dt = m.transformer
    dt.truncation_r = 0.85
    B = 32
    cond1 = synth.synth_cond_emb(1, key="p.cond")
    cond = cond1.expand(B, -1, -1).contiguous().cuda()
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    xt = synth.synth_tokens(1, mask_frac=0.5, key="p.xt").expand(B, -1).contiguous().cuda()
    t = torch.full((B,), 47, dtype=torch.long).cuda()
    u = synth.synth_uniform((1, 257, 265), key="p.u").expand(B, -1, -1).contiguous().cuda()
    a = dt.p_sample_tokens(xt, kv, t, u, initial=False).clone()
    b = dt.p_sample_tokens(xt, kv, t, u, initial=False).clone()
    assert torch.equal(a, b)                                   # deterministic
    assert (a == a[:1]).all()                                  # batch rows are independent and identical
    assert int(a.min()) >= 0 and int(a.max()) <= 256
    # a single-sample run must agree with row 0 of the batch (no cross-sample leakage through tiling)
    kv1 = dt.transformer.condition_kv(cond[:1].contiguous(), dt._schedule_table())
    one = dt.p_sample_tokens(xt[:1].contiguous(), kv1, t[:1].contiguous(), u[:1].contiguous(), initial=False)
    assert torch.equal(one, a[:1])
    # at t = 0 nothing can stay masked: q(x_{-1}) puts no mass on [MASK]
    t0 = torch.zeros(B, dtype=torch.long).cuda()
    z = dt.p_sample_tokens(xt, kv, t0, u, initial=False)
    assert int(z.max()) <= 255
    del m
    torch.cuda.empty_cache()
