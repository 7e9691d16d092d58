"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/Diffsound, via oracle/ref_harness.py) on seeded synthetic weights/inputs.

Run in the build container only:   python oracle/make_golden.py
The GPU box never runs this (no reference there); it consumes the committed vectors.
Inputs are not stored: they are re-derived from text-to-sound-synthesis_amd/synth.py by key.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from text_to_sound_synthesis_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
POS_STRIDE = 4          # per-step tensors are stored at every 4th grid position
STEP_CASES = ((99, None), (50, 0.55), (1, 0.02), (0, 0.0))   # (t, mask fraction of x_t)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


class InjectNoise:
    """Make the reference's torch.rand_like(logits) return our keyed noise."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        self.orig = torch.rand_like
        torch.rand_like = lambda x, *a, **k: self.fn(tuple(x.shape)).to(x.dtype)
        return self

    def __exit__(self, *a):
        torch.rand_like = self.orig


def state_keys(model, skip=()):
    """The state-dict contract of the boundary: parameter and buffer names/shapes."""
    ok = lambda k: not any(k.startswith(s) for s in skip)
    params = {k: list(v.shape) for k, v in model.named_parameters() if ok(k)}
    buffers = {k: list(v.shape) for k, v in model.state_dict().items() if ok(k) and k not in params}
    return {"params": params, "buffers": buffers}


@torch.no_grad()
def text_stage():
    # ---- (6) text stage: BPE tokens + CLIP text embedding (fp16 tower) -------------------------
    caps = synth.synth_captions(6) + ["It's 3 o'clock: RAIN &amp; thunder!!  (loud)   <café>", " ".join(["buzzing"] * 90)]
    tk = rh.reference_tokenize(caps)
    clip = rh.build_clip_text()
    emb = clip(tk["token"].clone())          # all 8 captions (round 4; rounds 1-3 stored the first two)
    with open(os.path.join(OUT, "captions.json"), "w") as f:
        json.dump(caps, f)
    save("text_stage", tokens=tk["token"], mask=tk["mask"], cond_emb=emb.float())
    with open(os.path.join(OUT, "state_dict_keys_clip.json"), "w") as f:
        json.dump({"transformer.condition_emb." + k: list(v.shape) for k, v in clip.state_dict().items()}, f,
                  indent=0, sort_keys=True)


@torch.no_grad()
def samplers():
    """top-k ('p') truncation, the skip-step sampler and the 'q' repeat-step sampler, each
    driven through the reference's own generate_content mini-language on the 2-layer T=10 model (B=2)."""
    import random
    torch.manual_seed(0)
    cond = synth.synth_cond_emb(2, key="traj.cond")
    arrs = {}

    # the reference's prepare_condition / decode are bypassed: call transformer.sample* the way generate_content does
    def drive(sample_type, key, seed=None):
        m = rh.build_dalle(n_layer=2, diffusion_step=10, n_embed=256)
        m.this_save_path = None
        dt = m.transformer
        parts = sample_type.split(",")
        if len(parts) > 1 and parts[1][:1] == "q":
            dt.p_sample = m.p_sample_with_truncation(dt.p_sample, parts[1])          # dalle_spec.py:205-206
        dt.predict_start = m.predict_start_with_truncation(dt.predict_start, parts[0])  # :207-209
        if seed is not None:
            random.seed(seed)
        n = [0]

        def noise(shp):
            n[0] += 1
            return synth.synth_uniform(shp, key="%s.u%d" % (key, n[0] - 1))
        with InjectNoise(noise):
            if len(parts) == 2 and parts[1][:4] == "fast":
                # sample_fast reads the batch size off condition_token (:769); its values are unused without CLIP
                out = dt.sample_fast(condition_token=torch.zeros(2, 77, dtype=torch.long), condition_mask=None,
                                     condition_embed=cond, filter_ratio=0, skip_step=int(parts[1][4:]))
            else:
                out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0,
                                batch_size=2)
        return m, out["content_token"], n[0]

    m, tok, n = drive("top100p", "topk")
    arrs.update(topk_tokens=tok, topk_calls=torch.tensor(n))
    # teacher-forced top-k step (T=10 model, t=5): the wrapper's output on a half-masked state
    from sound_synthesis.modeling.transformers.diffusion_transformer import index_to_log_onehot
    dt = m.transformer
    log_z = index_to_log_onehot(synth.synth_tokens(2, mask_frac=0.5, key="topk.xt"), 257)
    tvec = torch.tensor([5, 5])
    arrs.update(topk_trunc=dt.predict_start(log_z, cond, tvec)[:, :, ::POS_STRIDE])
    _, tok, n = drive("top0.85r,fast2", "fast")
    arrs.update(fast2_tokens=tok, fast2_calls=torch.tensor(n))
    _, tok, n = drive("top0.85r,q0.5", "rep", seed=7)
    arrs.update(q05_tokens=tok, q05_calls=torch.tensor(n))
    save("samplers_T10_L2", pos_stride=POS_STRIDE, **arrs)


def encoder():
    """VQModel.encode (Encoder + quant_conv + nearest-code search), DALLE.get_tokens, and
    sample()'s filter_ratio > 0 branch (q_sample of given tokens, then the reverse chain)."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=2, diffusion_step=10, n_embed=256, with_encoder=True)
    with open(os.path.join(OUT, "state_dict_keys_encoder.json"), "w") as f:
        keys = state_keys(m)
        keep = lambda k: k.startswith(("content_codec.encoder.", "content_codec.quant_conv."))
        json.dump({"encoder": {"params": {k: v for k, v in keys["params"].items() if keep(k)},
                               "buffers": {k: v for k, v in keys["buffers"].items() if keep(k)}}},
                  f, indent=0, sort_keys=True)
    mel = synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1
    codec = m.content_codec
    h = codec.quant_conv(codec.encoder(mel))
    quant, _, info = codec.encode(mel)
    _, tokens = m.get_tokens(mel)
    E = codec.quantize.embedding.weight
    z = h.permute(0, 2, 3, 1).reshape(-1, 256)
    d = (z ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * z @ E.t()
    top2 = d.topk(2, dim=1, largest=False).values
    arrs = dict(h=h, indices=info[2].view(2, -1), tokens=tokens, gap=(top2[:, 1] - top2[:, 0]).view(2, -1),
                quant_sample=quant[:, :, :, ::13])
    # partial re-sampling: the tokens above, filter_ratio 0.5 on the T=10 model -> q_sample at t=4, 5 reverse steps
    dt = m.transformer
    dt.predict_start = m.predict_start_with_truncation(dt.predict_start, "top0.85r")
    cond = synth.synth_cond_emb(2, key="traj.cond")
    n = [0]

    def noise(shp):
        n[0] += 1
        return synth.synth_uniform(shp, key="part.u%d" % (n[0] - 1))
    with InjectNoise(noise):
        out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, content_token=tokens,
                        filter_ratio=0.5, batch_size=2)
    arrs.update(partial_tokens=out["content_token"], partial_calls=torch.tensor(n[0]))
    save("encoder_T10_L2", **arrs)


def codebook512():
    """BASELINE configs[3] runs the 512-entry codebook (configs/caps_512.yaml:12,82 -> 513 classes): logits, one
    teacher-forced step and the decode of its tokens from the reference built with n_embed = 512."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=2, diffusion_step=100, n_embed=512)
    from sound_synthesis.modeling.transformers.diffusion_transformer import index_to_log_onehot
    dt = m.transformer
    keys = state_keys(m, skip=("content_codec.encoder.", "content_codec.quant_conv."))
    with open(os.path.join(OUT, "state_dict_keys_k512.json"), "w") as f:
        json.dump({"dalle_k512": keys}, f, indent=0, sort_keys=True)
    x = synth.synth_tokens(2, 265, 512, mask_frac=0.4, key="k512.x")
    cond = synth.synth_cond_emb(2, key="k512.c")
    t = torch.tensor([61, 12])
    logits = dt.transformer(x, cond, t)
    log_z = index_to_log_onehot(x, 513)
    log_pred = dt.predict_start(log_z, cond, t)
    trunc = m.predict_start_with_truncation(dt.predict_start, "top0.85r")(log_z, cond, t)
    post = dt.q_posterior(log_x_start=trunc, log_x_t=log_z, t=t)
    u = synth.synth_uniform((2, 513, 265), key="k512.u")
    with InjectNoise(lambda shp: u):
        toks = dt.log_sample_categorical(post).argmax(1)
    mel = m.decode_to_img(toks.clamp(max=511), (2, 256, 5, 53))
    s = slice(None, None, POS_STRIDE)
    save("k512_L2", pos_stride=POS_STRIDE, logits=logits[:, :, s], log_pred=log_pred[:, :, s], trunc=trunc[:, :, s],
         post=post[:, :, s], tokens=toks, kept=(trunc > -70).sum(1), mel0=mel[0])


def trained_like():
    """Off the N(0, 0.02) manifold (synth.py profile="trained": LayerNorm gains over two decades, hot residual-stream
    channels, heavy-tailed weights, one MLP unit driving GELU2 outputs into the 1e4s -- fp16 tops out at 65504): the
    19-layer denoiser's logits from the reference in fp32 (as shipped) AND in float64 (the same modules after .double():
    the yardstick that says how far fp32 itself is from the exact result on these weights), one teacher-forced step with
    injected noise, and the Gumbel-argmax margin of every decision of that step."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=19, diffusion_step=100, n_embed=256)
    synth.synth_init_(m, seed=0, skip=("content_codec.",), profile="trained")
    from sound_synthesis.modeling.transformers.diffusion_transformer import index_to_log_onehot
    dt = m.transformer
    tok = synth.synth_tokens(2, mask_frac=0.5, key="tl19.tokens")
    cond = synth.synth_cond_emb(2, key="tl19.cond")
    t = torch.tensor([63, 7])
    amax = {}

    def hook(name):
        def f(mod, inp, out):
            amax[name] = max(amax.get(name, 0.0), float(out.abs().max()))
        return f
    hs = []
    for blk in dt.transformer.blocks:
        hs += [blk.mlp[1].register_forward_hook(hook("gelu2_out")), blk.ln2.register_forward_hook(hook("ln2_out")),
               blk.register_forward_hook(lambda mod, inp, out: amax.__setitem__("block_out", max(amax.get("block_out", 0.0),
                                                                                 float(out[0].abs().max()))))]
    logits32 = dt.transformer(tok, cond, t)
    for h in hs:
        h.remove()
    # one teacher-forced step (B = 1) with the reference's truncation wrapper and injected noise
    x1 = synth.synth_tokens(1, mask_frac=0.55, key="tl19.step.xt")
    c1 = synth.synth_cond_emb(1, key="tl19.step.cond")
    tv = torch.tensor([50])
    log_z = index_to_log_onehot(x1, 257)
    trunc = m.predict_start_with_truncation(dt.predict_start, "top0.85r")(log_z, c1, tv)
    post = dt.q_posterior(log_x_start=trunc, log_x_t=log_z, t=tv)
    u = synth.synth_uniform((1, 257, 265), key="tl19.step.u")
    gum = -torch.log(-torch.log(u + 1e-30) + 1e-30) + post
    top2 = gum.topk(2, dim=1).values
    with InjectNoise(lambda shp: u):
        toks = dt.log_sample_categorical(post).argmax(1)
    logits64 = m.double().transformer.transformer(tok, cond.double(), t)
    s = slice(None, None, POS_STRIDE)
    save("transformer_L19_trainedlike", pos_stride=POS_STRIDE, logits=logits32[:, :, s], logits64=logits64[:, :, s],
         fp32_vs_fp64=float((logits32.double() - logits64).abs().max()), tokens=toks, margin=(top2[:, 0] - top2[:, 1]),
         kept=(trunc > -70).sum(1), amax_gelu2=amax["gelu2_out"], amax_ln2=amax["ln2_out"], amax_block=amax["block_out"])
    print("trained-like: logits |max| %.2f, reference fp32 vs float64 %.2e, GELU2 out |max| %.0f, residual |max| %.0f"
          % (float(logits64.abs().max()), float((logits32.double() - logits64).abs().max()), amax["gelu2_out"],
             amax["block_out"]))


def codebook512_L19():
    """The benchmarked K = 512 leg (BASELINE configs[3]) runs 19 layers: its logits from the reference (k512_L2.npz pins the
    2-layer build only)."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=19, diffusion_step=100, n_embed=512)
    x = synth.synth_tokens(2, 265, 512, mask_frac=0.4, key="k512.x")
    cond = synth.synth_cond_emb(2, key="k512.c")
    logits = m.transformer.transformer(x, cond, torch.tensor([61, 12]))
    save("k512_L19", pos_stride=POS_STRIDE, logits=logits[:, :, ::POS_STRIDE])


GRAD_PROBES = ("transformer.to_logits.1.weight", "transformer.to_logits.0.weight", "transformer.blocks.1.mlp.2.weight",
               "transformer.blocks.1.mlp.0.bias", "transformer.blocks.0.attn1.query.weight",
               "transformer.blocks.0.attn2.key.weight", "transformer.blocks.0.ln1.linear.weight",
               "transformer.blocks.0.ln1.emb.weight", "transformer.blocks.1.ln2.weight",
               "transformer.content_emb.emb.weight", "transformer.content_emb.width_emb.weight")


def train_loss():
    """SURVEY.md section 8f-3 (oracle groundwork): DiffusionTransformer.forward(return_loss=True) = _train_loss with
    the sampled timesteps and the q_sample noise injected (2-layer T=100 model, B=3, one sample at t = 0)."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=2, diffusion_step=100, n_embed=256)
    dt = m.transformer
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0")
    cond = synth.synth_cond_emb(3, key="tl.c")
    t = torch.tensor([57, 0, 93])
    pt = torch.ones(3) / 100
    dt.sample_time = lambda b, device, method="uniform": (t, pt)
    u = synth.synth_uniform((3, 257, 265), key="tl.u")
    with torch.enable_grad(), InjectNoise(lambda shp: u):
        out = dt({"content_token": x0, "condition_embed_token": cond, "condition_token": None}, return_loss=True,
                 return_logits=True)
    # gradients of that loss (what Solver.step back-propagates, engine/solver_spec.py): norms of a few parameters
    # spread over the network + the global norm
    params = dict(dt.named_parameters())
    for p_ in params.values():
        p_.requires_grad_(True)
        p_.grad = None
    Lt_h, Lt_c = dt.Lt_history.clone(), dt.Lt_count.clone()
    dt.Lt_history.zero_(); dt.Lt_count.zero_()
    with torch.enable_grad(), InjectNoise(lambda shp: u):
        out2 = dt({"content_token": x0, "condition_embed_token": cond, "condition_token": None}, return_loss=True)
        out2["loss"].backward()
    names = GRAD_PROBES
    gn = {n: params[n].grad.norm() for n in names}
    total = torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in params.values() if p_.grad is not None)).float()
    s = slice(None, None, POS_STRIDE)
    save("train_loss_L2", pos_stride=POS_STRIDE, loss=out["loss"].detach(), model_prob=out["logits"].detach()[:, :, s],
         Lt_history=Lt_h.detach(), Lt_count=Lt_c.detach(), grad_total=total,
         grad_logits_w_sample=params["transformer.to_logits.1.weight"].grad[::37, ::53].clone(),
         **{"gradnorm_" + n.replace(".", "_"): v for n, v in gn.items()})


TRAIN_BATCH_T = (57, 0, 93, 12, 99, 3, 71, 40, 88, 25, 64, 1, 97, 50, 33, 79, 8, 18, 45, 60)


def train_batch(name="train_batch_L19_b20", profile="init", n_layer=19, B=20):
    """Round 6: BASELINE configs[4] as the reference runs it -- the UNMODIFIED `DALLE.forward(batch,
    return_loss=True)` (dalle_spec.py:389-400 -> prepare_input :93-133: Tokenize + VQModel.encode of the mel; then
    diffusion_transformer.py:539-577: CLIPTextEmbedding of the caption ids, _train_loss) and `loss.backward()`
    (engine/solver_spec.py:308-331) at 19 layers, B = 20, on `image ~ U(-1, 1) f32[20,1,80,848]` + 20 synthetic captions
    (SURVEY.md section 8d row 5), with the sampled timesteps and the q_sample noise injected.  Stored: what every stage
    hands to the next (caption ids, CLIP embedding, VQ token ids + the argmin margin of every code), the loss, the norm
    of EVERY parameter gradient + the global norm, two gradient slices, and the importance-sampling statistics after."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=n_layer, diffusion_step=100, n_embed=256, with_encoder=True, with_clip=True)
    if profile != "init":
        synth.synth_init_(m, seed=0, skip=("content_codec.", "transformer.condition_emb."), profile=profile)
    dt = m.transformer
    mel = synth.synth_uniform((B, 1, 80, 848), key="tb.mel") * 2 - 1
    caps = synth.synth_captions(B, seed=17)
    t = torch.tensor(TRAIN_BATCH_T[:B])
    pt = torch.ones(B) / 100
    dt.sample_time = lambda b, device, method="uniform": (t, pt)
    u = synth.synth_uniform((B, 257, 265), key="tb.u")
    params = {k: p_ for k, p_ in dt.named_parameters() if not k.startswith("condition_emb.")}
    for p_ in params.values():
        p_.requires_grad_(True)
        p_.grad = None
    # what the stages hand over (recorded by calling the reference's own methods once more, outside the measured call)
    with torch.no_grad():
        inp = m.prepare_input({"image": mel, "text": caps})
        cond = dt.condition_emb(inp["condition_token"]).float()
        codec = m.content_codec
        h = codec.quant_conv(codec.encoder(mel))
        E = codec.quantize.embedding.weight
        zf = h.permute(0, 2, 3, 1).reshape(-1, 256)
        d = (zf ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * zf @ E.t()
        top2 = d.topk(2, dim=1, largest=False).values
        vq_gap = m.first_stage_permuter((top2[:, 1] - top2[:, 0]).view(B, -1))
    assert torch.equal(cond, cond.half().float())
    t0 = time.time()
    with torch.enable_grad(), InjectNoise(lambda shp: u):
        out = m({"image": mel, "text": caps}, return_loss=True)
        out["loss"].backward()
    print("reference DALLE.forward + backward, %d layers, B=%d, profile %s: %.1f s, loss %.6f"
          % (n_layer, B, profile, time.time() - t0, float(out["loss"])))
    names = sorted(k for k, p_ in params.items() if p_.grad is not None)
    norms = torch.stack([params[k].grad.double().norm() for k in names])
    total = torch.sqrt((norms ** 2).sum())
    amax = torch.stack([params[k].grad.abs().max() for k in names])
    with open(os.path.join(OUT, name + "_names.json"), "w") as f:
        json.dump({"captions": caps, "grad_names": names}, f)
    last = "transformer.blocks.%d." % (n_layer - 1)
    arrs = dict(t=t, caption_tokens=inp["condition_token"].to(torch.int32), cond_emb=cond.half(),
                tokens=inp["content_token"].to(torch.int16), vq_gap=vq_gap, loss=out["loss"].detach().double(),
                grad_norms=norms, grad_amax=amax, grad_total=total,
                grad_logits_w_sample=params["transformer.to_logits.1.weight"].grad[::37, ::53].clone(),
                grad_first_q_sample=params["transformer.blocks.0.attn1.query.weight"].grad[::97, ::89].clone(),
                grad_last_fc1_sample=params[last + "mlp.0.weight"].grad[::211, ::89].clone(),
                Lt_history=dt.Lt_history.detach().clone(), Lt_count=dt.Lt_count.detach().clone())
    # The yardstick's own error bar: the same loss + backward with the reference's modules in FLOAT64 (its own _train_loss,
    # :408-476, on the token ids and the CLIP embedding of the run above; forward() casts the embedding to fp32, :566, so the
    # method below it is called directly and the division of :570 repeated).  Some gradients are the result of a
    # cancellation -- d softmax of the near-uniform cross-attention of init-like weights: dS = P (dP - delta) with dP ~ delta
    # -- and the reference's fp32 resolves those to 1e-2 only; `grad_norms64` says, per tensor, how far fp32 is from exact.
    del out
    for p_ in params.values():
        p_.grad = None
    m.double()
    t0 = time.time()
    with torch.enable_grad(), InjectNoise(lambda shp: u):
        _, vb = dt._train_loss(inp["content_token"], cond.double())
        loss64 = vb.sum() / (inp["content_token"].size()[0] * inp["content_token"].size()[1])
        loss64.backward()
    print("  the same in float64: %.1f s, loss %.9f" % (time.time() - t0, float(loss64)))
    norms64 = torch.stack([params[k].grad.norm() for k in names])
    rel = ((norms - norms64).abs() / norms64.clamp(min=1e-30))
    worst = torch.argsort(rel, descending=True)[:4]
    print("  reference fp32 vs float64, worst per-tensor gradient norms: " +
          ", ".join("%s %.1e (|g| %.1e)" % (names[i], float(rel[i]), float(norms64[i])) for i in worst))
    save(name, loss64=loss64.detach(), grad_norms64=norms64, grad_total64=torch.sqrt((norms64 ** 2).sum()), **arrs)


def train_batch_trained():
    """The same iteration on trained-like denoiser weights (synth.py profile="trained"): the single 2^k loss scale and the
    saturation monitor meet a heavy-tailed dY."""
    train_batch("train_batch_L19_b20_trainedlike", profile="trained")


def solver_schedule():
    """SURVEY.md section 8f-3, the host logic around the training step (engine/solver_spec.py:308-331): the reference's
    own ReduceLROnPlateauWithWarmup, ClipGradNorm and EMA classes driven over a short deterministic run."""
    import types
    if "torch._six" not in sys.modules:          # removed from current torch; the reference imports `inf` from it
        m_ = types.ModuleType("torch._six"); m_.inf = float("inf"); sys.modules["torch._six"] = m_
    rh.install()
    from sound_synthesis.engine.lr_scheduler import ReduceLROnPlateauWithWarmup
    from sound_synthesis.engine.clip_grad_norm import ClipGradNorm
    from sound_synthesis.engine.ema import EMA
    torch.manual_seed(0)
    net = torch.nn.Linear(6, 4)
    with torch.no_grad():
        net.weight.copy_(synth.synth_uniform((4, 6), key="sv.w") - 0.5); net.bias.copy_(synth.synth_uniform((4,), key="sv.b") - 0.5)
    w0, b0 = net.weight.detach().clone(), net.bias.detach().clone()
    opt = torch.optim.AdamW(net.parameters(), lr=3.0e-6, betas=(0.9, 0.96), weight_decay=4.5e-2)
    sch = ReduceLROnPlateauWithWarmup(opt, factor=0.5, patience=4, min_lr=1.0e-6, threshold=1.0e-1, threshold_mode="rel",
                                      warmup_lr=4.5e-4, warmup=10)
    clip = ClipGradNorm(start_iteration=0, end_iteration=5, max_norm=0.5)
    ema = EMA(net, decay=0.99, update_interval=3)
    n = 40
    losses = synth.synth_uniform((n,), key="sv.loss") * 0.05 + torch.cat([torch.linspace(5, 1, 12), torch.ones(n - 12)])
    xs = synth.synth_uniform((n, 5, 6), key="sv.x") * 2 - 1
    lrs, gnorm_pre, gnorm_post = [], [], []
    for it in range(n):
        opt.zero_grad()
        out = (net(xs[it]) ** 2).sum() * 3.0
        out.backward()
        gnorm_pre.append(torch.sqrt(sum((p_.grad ** 2).sum() for p_ in net.parameters())))
        clip(net.parameters())
        gnorm_post.append(torch.sqrt(sum((p_.grad ** 2).sum() for p_ in net.parameters())))
        opt.step()
        sch.step(losses[it])
        ema.update(iteration=it)
        lrs.append(opt.param_groups[0]["lr"])
    es = ema.state_dict()
    save("solver_schedule", w0=w0, b0=b0, losses=losses, xs=xs, lrs=torch.tensor(lrs, dtype=torch.float64),
         gnorm_pre=torch.stack(gnorm_pre), gnorm_post=torch.stack(gnorm_post), w_final=net.weight.detach(),
         b_final=net.bias.detach(), ema_w=es["weight"], ema_b=es["bias"])


def dalle_sample():
    """DALLE.sample (the trainer's logging sampler, dalle_spec.py:264-343) on the T=10 two-layer model with the VQ
    encoder: reconstruction + re-sampling at filter_ratio 0 / 0.5 / 1.0 from the batch's own tokens.  The caption
    conditioning is injected as an embedding (prepare_condition stubbed: CLIP is pinned separately)."""
    torch.manual_seed(0)
    m = rh.build_dalle(n_layer=2, diffusion_step=10, n_embed=256, with_encoder=True)
    mel = synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1
    cond = synth.synth_cond_emb(2, key="traj.cond")
    m.prepare_condition = lambda batch, condition=None: {"condition_token": None, "condition_embed_token": cond}
    n = [0]

    def noise(shp):
        n[0] += 1
        return synth.synth_uniform(shp, key="ds.u%d" % (n[0] - 1))
    with InjectNoise(noise):
        out = m.sample({"image": mel, "text": ["a", "b"]}, filter_ratio=[0, 0.5, 1.0], content_ratio=[1], batch_size=2)
    s = slice(None, None, 7)
    save("dalle_sample_T10_L2", calls=torch.tensor(n[0]), time_stride=torch.tensor(7),
         reconstruction=out["reconstruction_image"][..., s], fr0=out["cond1_cont1_fr0_image"][..., s],
         fr05=out["cond1_cont1_fr0.5_image"][..., s], fr1=out["cond1_cont1_fr1.0_image"][..., s])


N1_B = 8                 # clips of the full-configuration trajectory golden
N1_WAVE_HEAD = 32768     # samples of each waveform that are stored


class _StopAfter(Exception):
    pass


N1_WAVE_FULL = (0, 7)   # clips whose waveform is stored at full length (fp32): the MelGAN tail of a sampled clip


def traj_full(name="traj_T100_L19", n_embed=256, n_clips=N1_B, caption_seed=11, n_steps=100, with_decode=True,
              profile="init"):
    """N1 (judge's row): end-to-end same-seed parity AT THE BENCHMARKED CONFIGURATION -- 19 layers, T = 100, K = 256,
    top0.85r, 8 captions.  Runs the reference's own loop (diffusion_transformer.py:587-659: 100 x p_sample :639-641)
    with the per-step noise injected, then dalle_spec.py:80-91 decode_to_img and vocoder/modules.py:129 forward.
    The conditioning is the reference's CLIPTextEmbedding of 8 synthetic captions (its rows are fp16 values, stored
    as fp16 without loss).  Besides the tokens after every step the file keeps, per (step, clip, position), the two
    quantities that decide whether a rounding-level logit difference can change a token:
      gap      top-1 minus top-2 of (gumbel + log posterior), the Gumbel-argmax margin      (:358-364)
      tmargin  min over classes of |mass ranked before the class - r|, the top-r cut margin (dalle_spec.py:160-173)
    so that the GPU test can demand that every disagreement sits on a near-tie.

    Round 5 variants of the same run (same hooks, same noise keys "n1.u<t>"):
      traj_T100_L19_k512   n_embed = 512 (configs/caps_512.yaml:12,82 -> 513 classes; BASELINE configs[3]), 8 captions, 100 steps
      traj_T100_L19_b64    64 DISTINCT captions, the first 10 of the 100 steps (the loop is left through an exception after
                           step t = 90; no decode): the benchmarked batch size against the reference itself, not 8 x 8 replicas"""
    torch.manual_seed(0)
    caps = synth.synth_captions(n_clips, seed=caption_seed)
    tk = rh.reference_tokenize(caps)
    clip = rh.build_clip_text()
    cond = clip(tk["token"].clone()).float()
    assert torch.equal(cond, cond.half().float())
    m = rh.build_dalle(n_layer=19, diffusion_step=100, n_embed=n_embed)
    if profile != "init":      # round 6: the denoiser off the N(0, 0.02) manifold (synth.py profile="trained"), as trained_like()
        synth.synth_init_(m, seed=0, skip=("content_codec.",), profile=profile)
    voc = rh.build_vocoder()
    dt = m.transformer
    inner = dt.predict_start
    r = 0.85
    tm = []

    def ps_with_margin(*a, **k):      # the reference wrapper's own arithmetic, plus the recorded cut margin
        out = inner(*a, **k)
        srt = torch.sort(out, 1, descending=True)[0]
        before = torch.exp(srt).cumsum(1)
        tm.append((before - r).abs().min(1)[0].clone())
        return out
    dt.predict_start = m.predict_start_with_truncation(ps_with_margin, "top0.85r")
    trace, gaps = [], []
    step = [99]
    orig_lsc = dt.log_sample_categorical

    def lsc(logits):
        u = synth.synth_uniform(tuple(logits.shape), key="n1.u%d" % step[0])
        g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
        top2 = (g + logits).topk(2, dim=1)[0]
        gaps.append((top2[:, 0] - top2[:, 1]).clone())
        with InjectNoise(lambda shp: u):
            out = orig_lsc(logits)
        trace.append(out.argmax(1).clone())
        step[0] -= 1
        if len(trace) == n_steps and n_steps < 100:
            raise _StopAfter()
        return out
    dt.log_sample_categorical = lsc
    t0 = time.time()
    try:
        out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0, batch_size=n_clips)
    except _StopAfter:
        out = None
    print("reference %d-step loop, B=%d, K=%d: %.1f s" % (n_steps, n_clips, n_embed, time.time() - t0))
    with open(os.path.join(OUT, name + "_captions.json"), "w") as f:
        json.dump(caps, f)
    common = dict(caption_tokens=tk["token"].to(torch.int32), cond_emb=cond.half(),
                  step_tokens=torch.stack(trace).to(torch.int16), gap=torch.stack(gaps).half(),
                  tmargin=torch.stack(tm).half())
    if out is None:
        assert not with_decode
        return save(name, **common)
    tokens = out["content_token"]
    assert step[0] == -1 and torch.equal(tokens, trace[-1])
    if not with_decode:
        return save(name, tokens=tokens.to(torch.int16), **common)
    mel = m.decode_to_img(tokens, (n_clips, 256, 5, 53))
    wave = voc((mel[:, 0] + 1) / 2)
    save(name, tokens=tokens.to(torch.int16), mel=mel[:, 0], wave_head=wave[:, 0, :N1_WAVE_HEAD],
         wave_full=wave[list(N1_WAVE_FULL), 0], wave_full_clips=torch.tensor(N1_WAVE_FULL), **common)


def traj_add_full_wave(name, n_embed):
    """Round 6: add `wave_full` (clips N1_WAVE_FULL, all 217 088 samples, fp32) to a chain golden written by an earlier
    round WITHOUT re-running its 100-step loop: the reference's decode_to_img + Generator.forward on the file's own final
    tokens.  The rebuilt models must reproduce the stored mel and waveform heads (to run-to-run rounding of torch-CPU), or nothing is written."""
    path = os.path.join(OUT, name + ".npz")
    z = dict(np.load(path))
    m = rh.build_dalle(n_layer=2, diffusion_step=100, n_embed=n_embed)      # (decode_to_img does not touch the denoiser)
    voc = rh.build_vocoder()
    tokens = torch.from_numpy(z["tokens"].astype(np.int64))
    with torch.no_grad():
        mel = m.decode_to_img(tokens, (tokens.shape[0], 256, 5, 53))
        wave = voc((mel[:, 0] + 1) / 2)
    # (not bit for bit: torch-CPU conv reductions depend on the thread count of the run; the two runs of the reference must
    # agree far inside the parity tolerances -- mel 1e-3 max-abs, waveform 1e-4 RMS -- or nothing is written)
    d_mel = float(np.abs(mel[:, 0].numpy() - z["mel"]).max())
    d_wav = float(np.sqrt(np.mean((wave[:, 0, :N1_WAVE_HEAD].numpy() - z["wave_head"]) ** 2)))
    print("re-run of the reference vs the stored vectors: mel max-abs %.2e, wave-head rms %.2e" % (d_mel, d_wav))
    assert d_mel < 2e-5 and d_wav < 2e-6, "rebuilt decoder / vocoder do not reproduce the stored vectors"
    z["wave_full"] = wave[list(N1_WAVE_FULL), 0].numpy()
    z["wave_full_clips"] = np.asarray(N1_WAVE_FULL)
    np.savez_compressed(path, **z)
    print("added wave_full to %-24s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def traj_trained():
    """Round 6: the chain golden on TRAINED-LIKE denoiser weights -- GELU2 outputs in the 1e4s, hot
    residual channels, LayerNorm gains over two decades, peaky posteriors whose top-r cut falls after rank 1-3 -- 8
    captions, 100 steps, same hooks / noise keys as traj_full."""
    traj_full("traj_T100_L19_trainedlike", n_embed=256, n_clips=8, caption_seed=11, n_steps=100, with_decode=True,
              profile="trained")


def traj_k512():
    """BASELINE configs[3] at chain level: the 512-key sort / 513-class tail over all 100 steps (8 captions, 19 layers)."""
    traj_full("traj_T100_L19_k512", n_embed=512, n_clips=8, caption_seed=11, n_steps=100, with_decode=True)


def traj_b64():
    """The benchmarked batch (64 distinct captions) against the reference: the first 10 reverse steps."""
    traj_full("traj_T100_L19_b64", n_embed=256, n_clips=64, caption_seed=13, n_steps=10, with_decode=False)


def signatures():
    """the Python signatures of the drop-in boundary, dumped from the reference itself.  The
    evaluation script is parsed (it imports soundfile / PIL / pandas at module level); the model classes are imported
    under the harness and read with inspect.  Each entry: [[name, default-or-null, kind], ...]."""
    import ast
    import inspect
    rh.install()
    out = {}

    def from_ast(path, cls, methods):
        tree = ast.parse(open(path).read())
        node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
        for fn in node.body:
            if isinstance(fn, ast.FunctionDef) and fn.name in methods:
                a = fn.args
                pos = a.posonlyargs + a.args
                defs = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
                out["%s.%s" % (cls, fn.name)] = [[x.arg, d, "POSITIONAL_OR_KEYWORD"] for x, d in zip(pos, defs)]

    def from_obj(name, fn):
        sig = inspect.signature(fn)
        out[name] = [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default), p.kind.name]
                     for p in sig.parameters.values()]
    from_ast(os.path.join(rh.REF_ROOT, "evaluation", "generate_samples_batch.py"), "Diffsound",
             ("__init__", "generate_sample", "inference_generate_sample_with_condition", "read_tsv"))
    from sound_synthesis.modeling.models.dalle_spec import DALLE
    from sound_synthesis.modeling.transformers.diffusion_transformer import DiffusionTransformer
    from sound_synthesis.modeling.transformers.transformer_utils import Text2ImageTransformer
    from sound_synthesis.modeling.codecs.spec_codec.vqgan import VQModel
    from specvqgan.modules.vqvae.quantize import VectorQuantizer
    from vocoder.modules import Generator
    for cls, names in ((DALLE, ("generate_content", "get_ema_model", "decode_to_img", "get_tokens", "prepare_condition",
                                "prepare_content", "forward", "sample")),
                       (DiffusionTransformer, ("sample", "sample_fast", "p_sample", "predict_start", "q_sample",
                                               "forward")),
                       (Text2ImageTransformer, ("forward",)),
                       (VQModel, ("decode", "encode", "forward")),
                       (VectorQuantizer, ("get_codebook_entry", "forward")),
                       (Generator, ("__init__", "forward"))):
        for n in names:
            f = getattr(cls, n)
            from_obj("%s.%s" % (cls.__name__, n), getattr(f, "__wrapped__", f))
    with open(os.path.join(OUT, "signatures.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote signatures.json (%d callables)" % len(out))


def bpe_closed_vocab():
    """text-to-sound-synthesis_amd/data/bpe_closed_vocab.json (package data, not a golden vector: it is input of the
    tokenizer, consumed by bench.py's timed region): the part of CLIP's merge table that the synthetic-caption word list
    (text_to_sound_synthesis_amd.synth._WORDS) exercises -- every merge the FULL table applies to one of those words, with
    its original rank, plus the ids of the resulting tokens and the two specials.  Checked here against the reference's
    own tokenizer on 3000 random captions over the word list.  (The GPU box has no copy of the full table.)"""
    from text_to_sound_synthesis_amd import synth as sy
    from text_to_sound_synthesis_amd import tokenizer as tz
    full = tz.SimpleTokenizer(bpe_path=os.path.join(rh.REF_ROOT, "sound_synthesis", "modeling", "modules", "clip",
                                                    "bpe_simple_vocab_16e6.txt.gz"))
    words = sorted(set(sy._WORDS))
    merges, tokens = set(), set()
    for w in words:
        syms = list(w[:-1]) + [w[-1] + "</w>"]
        while len(syms) > 1:                                # the tokenizer's own greedy loop, recording what it applies
            cand = [(full.rank[(a, b)], (a, b)) for a, b in zip(syms, syms[1:]) if (a, b) in full.rank]
            if not cand:
                break
            r, best = min(cand)
            merges.add((best[0], best[1], r))
            out, i = [], 0
            while i < len(syms):
                if i + 1 < len(syms) and (syms[i], syms[i + 1]) == best:
                    out.append(best[0] + best[1]); i += 2
                else:
                    out.append(syms[i]); i += 1
            syms = out
        assert syms == full._bpe(w)
        tokens.update(syms)
    enc = {t: full.encoder[t] for t in sorted(tokens)}
    for sp in ("<|startoftext|>", "<|endoftext|>"):
        enc[sp] = full.encoder[sp]
    path = tz.CLOSED_VOCAB_PATH
    with open(path, "w") as f:
        json.dump({"words": words, "merges": sorted(merges, key=lambda m: m[2]), "encoder": enc}, f)
    closed = tz.SimpleTokenizer(bpe_path=path)
    caps = sy.synth_captions(3000, seed=5)
    ref = rh.reference_tokenize(caps)["token"]
    got = tz.tokenize(caps, context_length=77, add_start_and_end=True, tokenizer=closed)["token"]
    assert torch.equal(ref, got)
    print("wrote bpe_closed_vocab.json (%d words, %d merges, %d tokens); 3000 captions equal the reference's ids"
          % (len(words), len(merges), len(enc)))


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--bpe-only" in sys.argv:
        return bpe_closed_vocab()
    if "--signatures-only" in sys.argv:
        return signatures()
    if "--n1-only" in sys.argv:
        return traj_full()
    if "--n1-k512-only" in sys.argv:
        return traj_k512()
    if "--n1-trained-only" in sys.argv:
        return traj_trained()
    if "--full-wave-only" in sys.argv:
        traj_add_full_wave("traj_T100_L19", 256)
        return traj_add_full_wave("traj_T100_L19_k512", 512)
    if "--train-batch-only" in sys.argv:
        return train_batch()
    if "--train-batch-trained-only" in sys.argv:
        return train_batch_trained()
    if "--train-batch-small" in sys.argv:      # plumbing check of the generator itself (not committed)
        return train_batch("_scratch_train_batch_L2_b3", n_layer=2, B=3)
    if "--n1-b64-only" in sys.argv:
        return traj_b64()
    if "--dsample-only" in sys.argv:
        return dalle_sample()
    if "--solver-only" in sys.argv:
        return solver_schedule()
    if "--text-only" in sys.argv:
        return text_stage()
    if "--train-only" in sys.argv:
        return train_loss()
    if "--k512-only" in sys.argv:
        return codebook512()
    if "--trained-only" in sys.argv:
        trained_like()
        return codebook512_L19()
    if "--encoder-only" in sys.argv:
        return encoder()
    if "--samplers-only" in sys.argv:
        return samplers()
    torch.manual_seed(0)
    t0 = time.time()

    # ---- (1) schedules + state-dict contract -------------------------------------------
    m2 = rh.build_dalle(n_layer=2, diffusion_step=100, n_embed=256)
    dt = m2.transformer
    names = ["log_at", "log_bt", "log_ct", "log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct",
             "log_1_min_ct", "log_1_min_cumprod_ct"]
    m10 = rh.build_dalle(n_layer=2, diffusion_step=10, n_embed=256)
    save("schedule", **{"T100_" + n: getattr(dt, n) for n in names},
         **{"T10_" + n: getattr(m10.transformer, n) for n in names})
    voc = rh.build_vocoder()
    m19 = rh.build_dalle(n_layer=19, diffusion_step=100, n_embed=256)
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump({"dalle": state_keys(m19, skip=("content_codec.encoder.", "content_codec.quant_conv.")),
                   "generator": state_keys(voc)}, f, indent=0, sort_keys=True)

    # ---- (2) transformer forward --------------------------------------------------------
    tok = synth.synth_tokens(2, mask_frac=0.3, key="tf2.tokens")
    cond = synth.synth_cond_emb(2, key="tf2.cond")
    t = torch.tensor([37, 80])
    save("transformer_L2", logits=dt.transformer(tok, cond, t))
    tok = synth.synth_tokens(1, mask_frac=0.5, key="tf19.tokens")
    cond1 = synth.synth_cond_emb(1, key="tf19.cond")
    save("transformer_L19", logits=m19.transformer.transformer(tok, cond1, torch.tensor([63])))
    del m19

    # ---- (3) teacher-forced single steps (2-layer model, T=100, B=1) --------------------
    trunc_fn = m2.predict_start_with_truncation(dt.predict_start, "top0.85r")
    from sound_synthesis.modeling.transformers.diffusion_transformer import index_to_log_onehot
    arrs = {}
    cond = synth.synth_cond_emb(1, key="step.cond")
    for tt, mf in STEP_CASES:
        tvec = torch.tensor([tt])
        if mf is None:   # the all-mask start state, diffusion_transformer.py:633-636
            log_z = torch.log(torch.cat((torch.zeros(1, 256, 265), torch.ones(1, 1, 265)), 1))
        else:
            log_z = index_to_log_onehot(synth.synth_tokens(1, mask_frac=mf, key="step%d.xt" % tt), 257)
        log_pred = dt.predict_start(log_z, cond, tvec)
        trunc = trunc_fn(log_z, cond, tvec)
        post = dt.q_posterior(log_x_start=trunc, log_x_t=log_z, t=tvec)
        u = synth.synth_uniform((1, 257, 265), key="step%d.u" % tt)
        with InjectNoise(lambda shp: u):
            toks = dt.log_sample_categorical(post).argmax(1)
        s = slice(None, None, POS_STRIDE)
        arrs.update({"t%d_log_pred" % tt: log_pred[:, :, s], "t%d_trunc" % tt: trunc[:, :, s],
                     "t%d_post" % tt: post[:, :, s], "t%d_tokens" % tt: toks,
                     "t%d_kept" % tt: (trunc > -70).sum(1)})
    save("steps_L2", pos_stride=POS_STRIDE, **arrs)

    # ---- (4) BASELINE config 1: 10-step trajectory -> decode -> vocoder (B=2) -----------
    d10 = m10.transformer
    d10.predict_start = m10.predict_start_with_truncation(d10.predict_start, "top0.85r")
    cond = synth.synth_cond_emb(2, key="traj.cond")
    trace = []
    orig_ps = d10.p_sample

    def traced(log_x, cond_emb, tvec):
        out = orig_ps(log_x, cond_emb, tvec)
        trace.append(out.argmax(1).clone())
        return out
    d10.p_sample = traced
    step_of_call = iter(range(9, -1, -1))
    with InjectNoise(lambda shp: synth.synth_uniform(shp, key="traj.u%d" % next(step_of_call))):
        out = d10.sample(condition_token=None, condition_mask=None, condition_embed=cond,
                         filter_ratio=0, batch_size=2)
    tokens = out["content_token"]
    mel = m10.decode_to_img(tokens, (2, 256, 5, 53))
    wave = voc((mel[:, 0] + 1) / 2)
    save("traj_T10_L2", step_tokens=torch.stack(trace), tokens=tokens, mel0=mel[0],
         wave0_head=wave[0, 0, :65536])

    # ---- (5) tokens -> mel (decoder) and mel -> wave (vocoder), B=1 ---------------------
    tok = synth.synth_tokens(1, mask_frac=0.0, key="dec.tokens")
    save("decode", mel=m2.decode_to_img(tok, (1, 256, 5, 53)))
    mel01 = synth.synth_uniform((1, 80, 848), key="voc.mel")
    save("vocoder", wave=voc(mel01))
    text_stage()
    samplers()
    encoder()
    codebook512()
    trained_like()
    codebook512_L19()
    train_loss()
    solver_schedule()
    dalle_sample()
    signatures()
    bpe_closed_vocab()
    traj_full()
    traj_k512()
    traj_b64()
    traj_trained()
    train_batch()
    train_batch_trained()
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
