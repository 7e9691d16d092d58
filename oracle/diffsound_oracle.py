"""TEST INFRASTRUCTURE -- the parity oracle.  Not part of the product.

A CPU restatement (plain torch-CPU / numpy, fp32 unless noted) of the reference's
Diffsound generation path, written as pure functions over a flat state dict whose
keys are the reference's own state-dict names.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this file; the product path
(text-to-sound-synthesis_amd/) never does and has no CPU fallback.

Pinning: the reference has no tests, golden vectors or checkpoints (SURVEY.md §4,
§8c), so this oracle is pinned against the reference *itself*: oracle/make_golden.py
(run where /root/reference exists) executes the unmodified reference modules on
seeded inputs and commits their outputs under tests/golden/; tests/test_oracle_golden.py
checks every function below against those vectors.

Every function cites the reference lines it restates (paths relative to
/root/reference/Diffsound/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_ZERO = float(np.log(np.float32(1e-30)))  # -69.0776 = log(1e-30), diffusion_transformer.py:53,306


# --------------------------------------------------------------------------- A11
def make_schedule(num_timesteps=100, num_classes=257):
    """Mask-and-uniform schedule buffers.
    sound_synthesis/modeling/transformers/diffusion_transformer.py:122-151 (alpha_schedule,
    att 0.99999->9e-6, ctt 9e-6->0.9, N = num_classes incl. [MASK]) and :193-231 (log
    buffers built in float64, then cast to float32)."""
    T, N = num_timesteps, num_classes
    lin = np.arange(0, T) / (T - 1)
    att = np.concatenate(([1.0], lin * (0.000009 - 0.99999) + 0.99999))
    ctt = np.concatenate(([0.0], lin * (0.9 - 0.000009) + 0.000009))
    at = att[1:] / att[:-1]
    ct = 1.0 - (1.0 - ctt[1:]) / (1.0 - ctt[:-1])
    bt = (1.0 - at - ct) / N
    att = np.concatenate((att[1:], [1.0]))
    ctt = np.concatenate((ctt[1:], [0.0]))
    btt = (1.0 - att - ctt) / N

    def lg(x):
        return torch.log(torch.tensor(x.astype("float64")))

    def l1m(la):  # log(1 - exp(a)), :25-26
        return torch.log(1 - la.exp() + 1e-40)

    log_ct, log_cct = lg(ct), lg(ctt)
    return {
        "log_at": lg(at).float(), "log_bt": lg(bt).float(), "log_ct": log_ct.float(),
        "log_cumprod_at": lg(att).float(), "log_cumprod_bt": lg(btt).float(),
        "log_cumprod_ct": log_cct.float(),
        "log_1_min_ct": l1m(log_ct).float(), "log_1_min_cumprod_ct": l1m(log_cct).float(),
    }


# --------------------------------------------------------------------------- A8
def content_embed(sd, tokens, pfx="transformer.transformer.content_emb.", hw=(5, 53)):
    """Token embedding + (row + column) position embedding.
    embeddings/dalle_mask_image_embedding.py:36-58: pos p uses height_emb[p // W] +
    width_emb[p % W] (row-major positions over the column-major token sequence)."""
    H, W = hw
    e = sd[pfx + "emb.weight"][tokens.clamp(min=0)]
    p = torch.arange(tokens.shape[1])
    pos = sd[pfx + "height_emb.weight"][p // W] + sd[pfx + "width_emb.weight"][p % W]
    return e + pos[None]


# --------------------------------------------------------------------------- A7
def _linear(sd, name, x):
    return x @ sd[name + ".weight"].t() + sd[name + ".bias"]


def _ada_ln(sd, name, x, t):
    """transformer_utils.py:134-149: LN(x) (no affine, eps 1e-5) * (1+scale) + shift with
    (scale, shift) = chunk(Linear(SiLU(Emb[t])))."""
    e = sd[name + ".emb.weight"][t]
    e = _linear(sd, name + ".linear", e * torch.sigmoid(e))[:, None, :]
    d = x.shape[-1]
    scale, shift = e[..., :d], e[..., d:]
    return F.layer_norm(x, (d,), eps=1e-5) * (1 + scale) + shift


def _mha(q, k, v, n_head):
    """softmax(q k^T / sqrt(hd)) v per head; transformer_utils.py:43-58 / :91-109
    (no mask, dropout p = 0)."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    hd = C // n_head
    qh = q.view(B, Lq, n_head, hd).transpose(1, 2)
    kh = k.view(B, Lk, n_head, hd).transpose(1, 2)
    vh = v.view(B, Lk, n_head, hd).transpose(1, 2)
    att = torch.softmax((qh @ kh.transpose(-2, -1)) * (1.0 / math.sqrt(hd)), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, Lq, C)


def transformer_block(sd, pfx, x, cond, t, n_head=16):
    """transformer_utils.py:255-272 ('selfcross'): x += attn1(ln1(x,t)); x += attn2(ln1_1(x,t),
    cond); x += mlp(ln2(x)); mlp = Linear, GELU2 (x*sigmoid(1.702x), :111-115), Linear."""
    h = _ada_ln(sd, pfx + "ln1", x, t)
    a = _mha(_linear(sd, pfx + "attn1.query", h), _linear(sd, pfx + "attn1.key", h),
             _linear(sd, pfx + "attn1.value", h), n_head)
    x = x + _linear(sd, pfx + "attn1.proj", a)
    h = _ada_ln(sd, pfx + "ln1_1", x, t)
    a = _mha(_linear(sd, pfx + "attn2.query", h), _linear(sd, pfx + "attn2.key", cond),
             _linear(sd, pfx + "attn2.value", cond), n_head)
    x = x + _linear(sd, pfx + "attn2.proj", a)
    h = F.layer_norm(x, (x.shape[-1],), sd[pfx + "ln2.weight"], sd[pfx + "ln2.bias"], eps=1e-5)
    h = _linear(sd, pfx + "mlp.0", h)
    h = h * torch.sigmoid(1.702 * h)
    return x + _linear(sd, pfx + "mlp.2", h)


def transformer_forward(sd, tokens, cond_emb, t, pfx="transformer.transformer.", n_head=16,
                        hw=(5, 53)):
    """Text2ImageTransformer.forward, transformer_utils.py:421-443 -> logits [B, K, L]."""
    x = content_embed(sd, tokens, pfx + "content_emb.", hw)
    i = 0
    while (pfx + "blocks.%d.ln2.weight" % i) in sd:
        x = transformer_block(sd, pfx + "blocks.%d." % i, x, cond_emb, t, n_head)
        i += 1
    x = F.layer_norm(x, (x.shape[-1],), sd[pfx + "to_logits.0.weight"],
                     sd[pfx + "to_logits.0.bias"], eps=1e-5)
    return _linear(sd, pfx + "to_logits.1", x).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- A6 / A3
def log_onehot(tokens, num_classes):
    """index_to_log_onehot, diffusion_transformer.py:45-56: log(clamp(onehot, 1e-30))."""
    oh = F.one_hot(tokens, num_classes).permute(0, 2, 1).float()
    return torch.log(oh.clamp(min=1e-30))


def initial_log_z(batch, num_classes=257, length=265):
    """All-[MASK] start state, diffusion_transformer.py:633-636: log([0,...,0,1]) = -inf/0."""
    z = torch.zeros(batch, num_classes, length)
    z[:, -1, :] = 1.0
    return torch.log(z)


def predict_start(logits):
    """diffusion_transformer.py:285-289: float64 log_softmax over classes, append a -70 row
    for [MASK], clamp to [-70, 0].  `logits` is the transformer output [B, K, L]."""
    lp = F.log_softmax(logits.double(), dim=1).float()
    lp = torch.cat((lp, torch.full_like(lp[:, :1, :], -70.0)), dim=1)
    return lp.clamp(-70.0, 0.0)


def truncate_top_r(log_pred, r):
    """Top-p ('r') truncation wrapper, models/dalle_spec.py:158-174.  Per column: rank the
    classes by descending log-prob; a class survives iff the probability mass ranked
    strictly before it is < r (rank 0 always survives); the rest become -70; no renorm."""
    srt, idx = torch.sort(log_pred, dim=1, descending=True)
    inc = torch.exp(srt).cumsum(dim=1)           # inclusive mass up to each rank
    keep_sorted = torch.cat((torch.ones_like(inc[:, :1, :], dtype=torch.bool),
                             (inc < r)[:, :-1, :]), dim=1)   # rank i looks at rank i-1's mass
    keep = torch.zeros_like(keep_sorted).scatter(1, idx, keep_sorted)
    return torch.where(keep, log_pred, torch.full_like(log_pred, -70.0))


def truncate_top_k(log_pred, k):
    """Top-k ('p') truncation wrapper, models/dalle_spec.py:147-157: the k largest log-probs of a column (over all
    K+1 rows) keep their value, every other row becomes -70."""
    val, ind = log_pred.topk(k=k, dim=1)
    return torch.full_like(log_pred, -70.0).scatter(1, ind, val)


# --------------------------------------------------------------------------- A9
def _lae(a, b):
    """log(exp a + exp b), diffusion_transformer.py:28-30."""
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def _q_pred(sched, log_x, t, T):
    """q(x_t | x_0) in log space, :253-267 (t wrapped mod T+1)."""
    t = (t + (T + 1)) % (T + 1)
    g = lambda n: sched[n][t].view(-1, 1, 1)
    return torch.cat((_lae(log_x[:, :-1] + g("log_cumprod_at"), g("log_cumprod_bt")),
                      _lae(log_x[:, -1:] + g("log_1_min_cumprod_ct"), g("log_cumprod_ct"))), dim=1)


def _q_pred_one(sched, log_x, t):
    """q(x_t | x_{t-1}) in log space, :241-251."""
    g = lambda n: sched[n][t].view(-1, 1, 1)
    return torch.cat((_lae(log_x[:, :-1] + g("log_at"), g("log_bt")),
                      _lae(log_x[:, -1:] + g("log_1_min_ct"), g("log_ct"))), dim=1)


def q_posterior(sched, log_x_start, log_x_t, t):
    """log p_theta(x_{t-1} | x_t), diffusion_transformer.py:293-339, clamped to [-70, 0]."""
    T = sched["log_at"].numel()
    Kp1 = log_x_start.shape[1]
    is_mask = (log_x_t.argmax(1) == Kp1 - 1).unsqueeze(1)           # [B,1,L]
    lz = torch.full_like(log_x_t[:, :1, :], LOG_ZERO)

    def fix(log_q, per_t):
        # mask row -> log(1e-30); where x_t is [MASK]: classes -> per_t, mask row -> 0
        log_q = torch.cat((log_q[:, :-1], lz), dim=1)
        alt = torch.cat((per_t.view(-1, 1, 1).expand(-1, Kp1 - 1, 1),
                         torch.zeros(log_q.shape[0], 1, 1)), dim=1)
        return torch.where(is_mask, alt.expand_as(log_q), log_q)

    log_qt = fix(_q_pred(sched, log_x_t, t, T), sched["log_cumprod_ct"][t])
    log_q1 = fix(_q_pred_one(sched, log_x_t, t), sched["log_ct"][t])
    q = log_x_start - log_qt
    lse = torch.logsumexp(q, dim=1, keepdim=True)
    q = q - lse
    out = _q_pred(sched, q, t - 1, T) + log_q1 + lse
    return out.clamp(-70.0, 0.0)


# --------------------------------------------------------------------------- A10
def gumbel_sample(log_prob, u):
    """log_sample_categorical, :359-368, with the uniform noise `u` injected
    (the reference draws torch.rand_like(logits)).  Returns token ids [B, L]."""
    g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    return (g + log_prob).argmax(dim=1)


# --------------------------------------------------------------------------- A4 / A5
def p_sample_step(sd, sched, log_z, cond_emb, t, u, trunc_r=0.85, n_head=16, detail=False, trunc_k=None,
                  t_post=None):
    """One reverse step, :342-357 with the truncation wrapper of dalle_spec.py:208-210
    installed: argmax -> transformer -> log-softmax(f64) -> top-r (or top-k) -> posterior -> Gumbel.
    t_post: timestep of the posterior when it differs from the network's (sample_fast, :796-803)."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    x_t = log_z.argmax(1)
    logits = transformer_forward(sd, x_t, cond_emb, t, n_head=n_head)
    log_pred = predict_start(logits)
    if trunc_k is not None:
        trunc = truncate_top_k(log_pred, trunc_k)
    else:
        trunc = truncate_top_r(log_pred, trunc_r) if trunc_r is not None else log_pred
    post = q_posterior(sched, trunc, log_z, t if t_post is None else t_post)
    tok = gumbel_sample(post, u)
    new_log_z = log_onehot(tok, K + 1)
    if detail:
        return new_log_z, dict(x_t=x_t, logits=logits, log_pred=log_pred, trunc=trunc,
                               post=post, tokens=tok)
    return new_log_z


def sample_loop(sd, cond_emb, noise_fn, num_timesteps=100, trunc_r=0.85, n_head=16,
                record=None):
    """DiffusionTransformer.sample with filter_ratio = 0, :633-641,654.
    noise_fn(step_t, shape) supplies the uniform noise for step t."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    L = 265
    B = cond_emb.shape[0]
    sched = make_schedule(num_timesteps, K + 1)
    log_z = initial_log_z(B, K + 1, L)
    for step in range(num_timesteps - 1, -1, -1):
        t = torch.full((B,), step, dtype=torch.long)
        log_z = p_sample_step(sd, sched, log_z, cond_emb, t, noise_fn(step, log_z.shape),
                              trunc_r, n_head)
        if record is not None:
            record.append(log_z.argmax(1).clone())
    return log_z.argmax(1)


def sample_loop_fast(sd, cond_emb, noise_fn, skip_step, num_timesteps=100, trunc_r=0.85, n_head=16, record=None):
    """DiffusionTransformer.sample_fast, :748-812: timesteps T-1, T-2-skip, ... with 0 appended; the network sees
    t, q_posterior sees t - skip_step while t > skip_step.  noise_fn(step_t, shape) as in sample_loop."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    B = cond_emb.shape[0]
    sched = make_schedule(num_timesteps, K + 1)
    log_z = initial_log_z(B, K + 1, 265)
    steps = list(range(num_timesteps - 1, -1, -1 - skip_step))
    if steps[-1] != 0:
        steps.append(0)
    for step in steps:
        t = torch.full((B,), step, dtype=torch.long)
        log_z = p_sample_step(sd, sched, log_z, cond_emb, t, noise_fn(step, log_z.shape), trunc_r, n_head,
                              t_post=t - skip_step if step > skip_step else t)
        if record is not None:
            record.append(log_z.argmax(1).clone())
    return log_z.argmax(1)


def sample_loop_repeat(sd, cond_emb, noise_fn, rate, rng, num_timesteps=100, trunc_r=0.85, n_head=16, record=None):
    """sample() with the 'q' wrapper on p_sample, dalle_spec.py:135-143: after every step, with probability `rate`
    (one rng.random() per step, Python's `random` in the reference) the step is applied again at the same t.
    noise_fn(call_index, shape): one draw per p_sample call."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    B = cond_emb.shape[0]
    sched = make_schedule(num_timesteps, K + 1)
    log_z = initial_log_z(B, K + 1, 265)
    calls = 0
    for step in range(num_timesteps - 1, -1, -1):
        t = torch.full((B,), step, dtype=torch.long)
        log_z = p_sample_step(sd, sched, log_z, cond_emb, t, noise_fn(calls, log_z.shape), trunc_r, n_head)
        calls += 1
        if rng.random() < rate:
            log_z = p_sample_step(sd, sched, log_z, cond_emb, t, noise_fn(calls, log_z.shape), trunc_r, n_head)
            calls += 1
        if record is not None:
            record.append(log_z.argmax(1).clone())
    return log_z.argmax(1)


# --------------------------------------------------------------------------- scope row 8f-3 (oracle only so far)
def train_loss(sd, x0, cond_emb, t, pt, u, num_timesteps=100, n_head=16, mask_weight=(1.0, 1.0),
               auxiliary_loss_weight=5.0e-4, adaptive_auxiliary_loss=True):
    """DiffusionTransformer._train_loss + the normalisation of forward(), diffusion_transformer.py:408-476,571-574:
    x_t ~ q(x_t | x_0) (uniforms u injected), the network's p(x_0 | x_t), the KL between the true and the modelled
    posterior (decoder NLL at t = 0), re-weighted by 1/pt, plus the auxiliary KL(x_0 || p(x_0|x_t)) term.
    x0 i64[B, L] tokens, t i64[B] and pt f32[B] as sample_time() returned them.  No truncation wrapper is installed
    in training.  Returns (log_model_prob [B, K+1, L], vb_loss [B], loss scalar, Lt2 [B] = kl_loss^2 for Lt_history)."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    sched = make_schedule(num_timesteps, K + 1)
    log_x_start = log_onehot(x0, K + 1)
    log_xt = q_sample(sched, x0, t, u, K + 1)
    xt = log_xt.argmax(1)
    log_x0_recon = predict_start(transformer_forward(sd, xt, cond_emb, t, n_head=n_head))
    log_model_prob = q_posterior(sched, log_x0_recon, log_xt, t)
    log_true_prob = q_posterior(sched, log_x_start, log_xt, t)
    kl_of = lambda a, b: (a.exp() * (a - b)).sum(dim=1)                       # multinomial_kl, :237-239
    mask_region = (xt == K).float()
    weight = mask_region * mask_weight[0] + (1.0 - mask_region) * mask_weight[1]
    kl = (kl_of(log_true_prob, log_model_prob) * weight).sum(-1)
    decoder_nll = -(log_x_start.exp() * log_model_prob).sum(dim=1).sum(-1)    # -log_categorical, :42-43
    is0 = (t == 0).float()
    kl_loss = is0 * decoder_nll + (1.0 - is0) * kl
    vb_loss = kl_loss / pt
    if auxiliary_loss_weight != 0:
        kl_aux = (kl_of(log_x_start[:, :-1, :], log_x0_recon[:, :-1, :]) * weight).sum(-1)
        kl_aux_loss = is0 * decoder_nll + (1.0 - is0) * kl_aux
        w = t.float() / num_timesteps + 1.0 if adaptive_auxiliary_loss else 1.0
        vb_loss = vb_loss + w * auxiliary_loss_weight * kl_aux_loss / pt
    loss = vb_loss.sum() / (x0.shape[0] * x0.shape[1])
    return log_model_prob, vb_loss, loss, kl_loss.pow(2)


def loss_tail_backward(sched, logits, x0, xt, t, pt, num_timesteps=100, mask_weight=(1.0, 1.0),
                       auxiliary_loss_weight=5.0e-4, adaptive_auxiliary_loss=True):
    """Closed-form d(sum_b vb_loss_b) / d logits of train_loss() -- the formula a HIP loss-tail backward kernel will
    implement (plan for row 8f-3); tests compare it with autograd through the forward restatement.
    logits f32[B, K, L] (the network output at (x_t, t)); returns the gradient, same shape."""
    B, K, L = logits.shape
    T = num_timesteps
    tt = (t + (T + 1)) % (T + 1)
    tm1 = (t - 1 + (T + 1)) % (T + 1)
    g = lambda n, idx: sched[n][idx].view(-1, 1, 1)
    lsm = torch.log_softmax(logits.double(), dim=1).float()
    lp = lsm.clamp(-70.0, 0.0)
    lp_live = (lsm > -70.0) & (lsm < 0.0)                                    # where the clamp passes gradient
    log_xt, log_x0 = log_onehot(xt, K + 1), log_onehot(x0, K + 1)
    is_mask = (xt == K).unsqueeze(1)
    # log q(x_t | x_0 = c) per row c (the `fix`ed q_pred of q_posterior) and the one-step version
    qt_cls = torch.where(is_mask, g("log_cumprod_ct", tt).expand(B, K, L),
                         _lae(log_xt[:, :-1] + g("log_cumprod_at", tt), g("log_cumprod_bt", tt)))
    qt_m = torch.where(is_mask, torch.zeros(B, 1, L), torch.full((B, 1, L), LOG_ZERO))
    q1_cls = torch.where(is_mask, g("log_ct", t).expand(B, K, L), _lae(log_xt[:, :-1] + g("log_at", t), g("log_bt", t)))
    q1_m = qt_m

    def posterior(lx0_cls, lx0_m):
        q = torch.cat((lx0_cls - qt_cls, lx0_m - qt_m), dim=1)
        lse = torch.logsumexp(q, dim=1, keepdim=True)
        a = torch.cat((g("log_cumprod_at", tm1).expand(B, K, 1), g("log_1_min_cumprod_ct", tm1)), dim=1)
        b = torch.cat((g("log_cumprod_bt", tm1).expand(B, K, 1), g("log_cumprod_ct", tm1)), dim=1)
        A = _lae(q - lse + a, b)
        raw = A + torch.cat((q1_cls, q1_m), dim=1) + lse
        sigma = torch.exp(q - lse + a - A)
        return raw.clamp(-70.0, 0.0), (raw > -70.0) & (raw < 0.0), sigma, torch.exp(q - lse)

    pm, pm_live, sigma, p = posterior(lp, torch.full((B, 1, L), -70.0))
    pr, _, _, _ = posterior(log_x0[:, :-1], log_x0[:, -1:])
    mask_region = (xt == K).float().unsqueeze(1)
    weight = mask_region * mask_weight[0] + (1.0 - mask_region) * mask_weight[1]         # [B,1,L]
    is0 = (t == 0).float().view(-1, 1, 1)
    ipt = (1.0 / pt).view(-1, 1, 1)
    # upstream weights on the model posterior rows: KL term (t > 0) and decoder NLL (t == 0, in both loss terms)
    nll_scale = ipt
    if auxiliary_loss_weight != 0:
        wa = (t.float() / T + 1.0 if adaptive_auxiliary_loss else torch.ones_like(t, dtype=torch.float32)).view(-1, 1, 1)
        nll_scale = ipt + wa * auxiliary_loss_weight * ipt
    w = (-(1.0 - is0) * torch.exp(pr) * weight * ipt - is0 * torch.exp(log_x0) * nll_scale) * pm_live
    s = (w * (1.0 - sigma)).sum(dim=1, keepdim=True)
    g_q = w * sigma + p * s                                                   # over the K+1 rows
    g_lp = g_q[:, :-1]                                                        # the [MASK] row of lp is a constant
    if auxiliary_loss_weight != 0:
        g_lp = g_lp - (1.0 - is0) * wa * auxiliary_loss_weight * ipt * weight * torch.exp(log_x0[:, :-1])
    g_lp = g_lp * lp_live
    return g_lp - torch.softmax(logits.double(), dim=1).float() * g_lp.sum(dim=1, keepdim=True)


# --------------------------------------------------------------------------- A12
def codebook_gather(sd, tokens, hw=(5, 53), pfx="content_codec."):
    """decode_to_img's first half, dalle_spec.py:80-89: ColumnMajor reverse permutation
    (permuter.py:31-55: row-major cell (h,w) takes sequence item w*H + h) followed by the
    codebook lookup (quantize.py:88-103) -> [B, C, H, W]."""
    H, W = hw
    hh = torch.arange(H).view(H, 1)
    ww = torch.arange(W).view(1, W)
    src = (ww * H + hh).reshape(-1)                                  # row-major -> seq index
    z = sd[pfx + "quantize.embedding.weight"][tokens[:, src]]       # [B, H*W, C]
    return z.view(tokens.shape[0], H, W, -1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------- A13 / A14
def _gn_swish(sd, name, x):
    """Normalize = GroupNorm(32, C, eps 1e-6) (model.py:34-35) then swish (:29-31)."""
    h = F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)
    return h * torch.sigmoid(h)


def _conv2d(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def _res_block(sd, pfx, x):
    """ResnetBlock.forward with temb=None, model.py:131-151."""
    h = _conv2d(sd, pfx + "conv1", _gn_swish(sd, pfx + "norm1", x), 1)
    h = _conv2d(sd, pfx + "conv2", _gn_swish(sd, pfx + "norm2", h), 1)
    if (pfx + "nin_shortcut.weight") in sd:
        x = _conv2d(sd, pfx + "nin_shortcut", x, 0)
    return x + h


def _attn_block(sd, pfx, x):
    """AttnBlock.forward, model.py:202-226: single head over H*W positions, scale C^-0.5."""
    B, C, H, W = x.shape
    h = F.group_norm(x, 32, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], eps=1e-6)
    q = _conv2d(sd, pfx + "q", h, 0).reshape(B, C, H * W)
    k = _conv2d(sd, pfx + "k", h, 0).reshape(B, C, H * W)
    v = _conv2d(sd, pfx + "v", h, 0).reshape(B, C, H * W)
    w = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (int(C) ** (-0.5)), dim=2)  # [B, i, j]
    o = torch.bmm(v, w.transpose(1, 2)).reshape(B, C, H, W)
    return x + _conv2d(sd, pfx + "proj_out", o, 0)


def vq_decode(sd, quant, pfx="content_codec.", num_resolutions=5, num_res_blocks=2,
              taps=None):
    """VQModel.decode (spec_codec/vqgan.py:62-65) = post_quant_conv + Decoder.forward
    (specvqgan/modules/diffusionmodules/model.py:640-671).  `taps` (a dict) receives
    intermediate activations for debugging the HIP path."""
    d = pfx + "decoder."
    h = _conv2d(sd, pfx + "post_quant_conv", quant, 0)
    h = _conv2d(sd, d + "conv_in", h, 1)
    h = _res_block(sd, d + "mid.block_1.", h)
    h = _attn_block(sd, d + "mid.attn_1.", h)
    h = _res_block(sd, d + "mid.block_2.", h)
    if taps is not None:
        taps["mid"] = h
    for lvl in reversed(range(num_resolutions)):
        for ib in range(num_res_blocks + 1):
            h = _res_block(sd, d + "up.%d.block.%d." % (lvl, ib), h)
            if (d + "up.%d.attn.%d.norm.weight" % (lvl, ib)) in sd:
                h = _attn_block(sd, d + "up.%d.attn.%d." % (lvl, ib), h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")    # Upsample, model.py:48-52
            h = _conv2d(sd, d + "up.%d.upsample.conv" % lvl, h, 1)
        if taps is not None:
            taps["up%d" % lvl] = h
    h = _gn_swish(sd, d + "norm_out", h)
    return _conv2d(sd, d + "conv_out", h, 1)


def decode_tokens(sd, tokens, pfx="content_codec."):
    """DALLE.decode_to_img, dalle_spec.py:80-91 -> mel-like image [B, 1, 80, 848]."""
    return vq_decode(sd, codebook_gather(sd, tokens, pfx=pfx), pfx=pfx)


# --------------------------------------------------------------------------- scope row 8f-2
def vq_encoder(sd, x, pfx="content_codec.", num_resolutions=5, num_res_blocks=2):
    """Encoder.forward (specvqgan/modules/diffusionmodules/model.py:467-500) + quant_conv
    (spec_codec/vqgan.py:54-56): mel image [B, 1, 80, 848] -> pre-quantisation latent [B, 256, 5, 53].
    Downsample = zero pad (0,1,0,1) then 3x3 stride-2 conv (:60-77); attention only where the blocks carry it
    (the 53-wide level)."""
    e = pfx + "encoder."
    h = _conv2d(sd, e + "conv_in", x, 1)
    for lvl in range(num_resolutions):
        for ib in range(num_res_blocks):
            h = _res_block(sd, e + "down.%d.block.%d." % (lvl, ib), h)
            if (e + "down.%d.attn.%d.norm.weight" % (lvl, ib)) in sd:
                h = _attn_block(sd, e + "down.%d.attn.%d." % (lvl, ib), h)
        if lvl != num_resolutions - 1:
            name = e + "down.%d.downsample.conv" % lvl
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[name + ".weight"], sd[name + ".bias"], stride=2)
    h = _res_block(sd, e + "mid.block_1.", h)
    h = _attn_block(sd, e + "mid.attn_1.", h)
    h = _res_block(sd, e + "mid.block_2.", h)
    h = _conv2d(sd, e + "conv_out", _gn_swish(sd, e + "norm_out", h), 1)
    return _conv2d(sd, pfx + "quant_conv", h, 0)


def vq_quantize(sd, h, pfx="content_codec."):
    """VectorQuantizer.forward's code search, vqvae/quantize.py:31-53: rows of h (channels last) against the
    codebook with d = |z|^2 + |e|^2 - 2 z.e, argmin.  Returns (indices [B, H*W] row-major, d [B*H*W, K])."""
    E = sd[pfx + "quantize.embedding.weight"]
    z = h.permute(0, 2, 3, 1).contiguous().view(-1, E.shape[1])
    d = torch.sum(z ** 2, dim=1, keepdim=True) + torch.sum(E ** 2, dim=1) - 2 * torch.matmul(z, E.t())
    return torch.argmin(d, dim=1).view(h.shape[0], -1), d


def encode_tokens(sd, mel, pfx="content_codec.", hw=(5, 53)):
    """DALLE.get_tokens, dalle_spec.py:70-77: encode -> indices -> ColumnMajor (sequence item w*H + h is cell (h, w),
    permuter.py:21-55)."""
    idx, _ = vq_quantize(sd, vq_encoder(sd, mel, pfx), pfx)
    H, W = hw
    return idx.view(-1, H, W).transpose(1, 2).reshape(idx.shape[0], H * W)


def q_sample(sched, tokens, t, u, num_classes):
    """q_sample, diffusion_transformer.py:370-377: x_t ~ q(x_t | x_0) via q_pred (:253-267) and the Gumbel
    sampler with injected uniforms u [B, K+1, L].  Returns the log-one-hot state."""
    T = sched["log_at"].numel()
    log_x0 = log_onehot(tokens, num_classes)
    return log_onehot(gumbel_sample(_q_pred(sched, log_x0, t, T), u), num_classes)


def sample_loop_partial(sd, cond_emb, content_token, filter_ratio, noise_fn, num_timesteps=100, trunc_r=0.85,
                        n_head=16):
    """sample() with filter_ratio > 0, :643-651: diffuse the given tokens forward to t = start_step - 1, then run the
    reverse chain from there.  noise_fn(call_index, shape): call 0 is q_sample's draw."""
    K = sd["transformer.transformer.to_logits.1.weight"].shape[0]
    B = cond_emb.shape[0]
    sched = make_schedule(num_timesteps, K + 1)
    start = int(num_timesteps * filter_ratio)
    t = torch.full((B,), start - 1, dtype=torch.long)
    log_z = q_sample(sched, content_token, t, noise_fn(0, (B, K + 1, content_token.shape[1])), K + 1)
    calls = 1
    for step in range(start - 1, -1, -1):
        t = torch.full((B,), step, dtype=torch.long)
        log_z = p_sample_step(sd, sched, log_z, cond_emb, t, noise_fn(calls, log_z.shape), trunc_r, n_head)
        calls += 1
    return log_z.argmax(1)


# --------------------------------------------------------------------------- A15
def _wn(sd, name):
    """weight_norm fold: w = g * v / ||v||, norm over all dims but 0 (vocoder/modules.py:18-23;
    for ConvTranspose1d dim 0 is the *input* channel)."""
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


def melgan_generator(sd, mel, pfx="model.", ratios=(8, 8, 2, 2), n_res=3):
    """Generator.forward, vocoder/modules.py:88-130.  mel [B, 80, T] in [0,1] ->
    waveform [B, 1, 256*T]."""
    lrelu = lambda x: F.leaky_relu(x, 0.2)
    i = 1
    x = F.conv1d(F.pad(mel, (3, 3), mode="reflect"), _wn(sd, pfx + "1"), sd[pfx + "1.bias"])
    i = 2
    for r in ratios:
        # LeakyReLU at i, WNConvTranspose1d at i+1: k=2r, s=r, p=r//2+r%2, out_pad=r%2
        x = F.conv_transpose1d(lrelu(x), _wn(sd, pfx + "%d" % (i + 1)), sd[pfx + "%d.bias" % (i + 1)],
                               stride=r, padding=r // 2 + r % 2, output_padding=r % 2)
        i += 2
        for j in range(n_res):
            b = pfx + "%d." % i
            dil = 3 ** j
            h = F.pad(lrelu(x), (dil, dil), mode="reflect")
            h = F.conv1d(h, _wn(sd, b + "block.2"), sd[b + "block.2.bias"], dilation=dil)
            h = F.conv1d(lrelu(h), _wn(sd, b + "block.4"), sd[b + "block.4.bias"])
            x = F.conv1d(x, _wn(sd, b + "shortcut"), sd[b + "shortcut.bias"]) + h
            i += 1
    x = F.pad(lrelu(x), (3, 3), mode="reflect")
    x = F.conv1d(x, _wn(sd, pfx + "%d" % (i + 2)), sd[pfx + "%d.bias" % (i + 2)])
    return torch.tanh(x)


def mel_to_unit(x):
    """generate_samples_batch.py:181-182: spec = (x + 1) / 2 before saving / vocoding."""
    return (x + 1.0) / 2.0


# --------------------------------------------------------------------------- A16 (scope row 8f-1)
def clip_text_embed(sd, tokens, pfx="transformer.condition_emb.", heads=8):
    """CLIPTextEmbedding.forward with pick_last_embedding False (embeddings/clip_text_embedding.py:46-88):
    the ViT-B/32 text tower of modules/clip/model.py (ResidualAttentionBlock :166-186, causal mask
    :313-319, QuickGELU :160-162) run the way the reference runs it: Linear / MHA weights in fp16
    (convert_weights :373-395), activations fp16, LayerNorm computed in fp32 (:150-157)."""
    d = sd[pfx + "ln_final.weight"].shape[0]
    h = lambda k: sd[pfx + k].half()
    ln = lambda x, n: F.layer_norm(x.float(), (d,), sd[pfx + n + ".weight"].float(), sd[pfx + n + ".bias"].float(),
                                   1e-5).half()
    tok = tokens.clamp(min=0)
    x = sd[pfx + "token_embedding.weight"][tok].half() + sd[pfx + "positional_embedding"].half()
    x = x.permute(1, 0, 2)                                            # [L, B, D]
    Ls = x.shape[0]
    mask = torch.full((Ls, Ls), float("-inf")).triu_(1).half()
    i = 0
    while (pfx + "transformer.resblocks.%d.ln_1.weight" % i) in sd:
        b = "transformer.resblocks.%d." % i
        y = ln(x, b + "ln_1")
        a = F.multi_head_attention_forward(
            y, y, y, d, heads, h(b + "attn.in_proj_weight"), h(b + "attn.in_proj_bias"), None, None, False, 0.0,
            h(b + "attn.out_proj.weight"), h(b + "attn.out_proj.bias"), training=False, need_weights=False,
            attn_mask=mask)[0]
        x = x + a
        y = F.linear(ln(x, b + "ln_2"), h(b + "mlp.c_fc.weight"), h(b + "mlp.c_fc.bias"))
        y = y * torch.sigmoid(1.702 * y)
        x = x + F.linear(y, h(b + "mlp.c_proj.weight"), h(b + "mlp.c_proj.bias"))
        i += 1
    x = ln(x.permute(1, 0, 2), "ln_final")
    x = x / x.norm(dim=-1, keepdim=True)
    return x.float()
