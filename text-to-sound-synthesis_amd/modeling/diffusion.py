"""Drop-in for sound_synthesis/modeling/transformers/diffusion_transformer.py:DiffusionTransformer
(sampling side), HIP-backed.

State-dict keys (log_at ... Lt_count, transformer.*), constructor keywords and the
sample()/p_sample()/predict_start() signatures follow the reference (:153-234, :269-291, :342-357,
:587-659).  The reverse loop carries token indices; one step = ds_denoiser_step (19-block denoiser +
the fused predict_start / top-r truncation / q_posterior / Gumbel-argmax tail).  The uniform noise is
drawn with torch.rand on the reference's [B, K+1, L] shape so that the same seed consumes the same
Philox stream as the reference's torch.rand_like(logits) (:360).
"""
import numpy as np

import torch
from torch import nn

from .. import _lib
from ..config import instantiate_from_config


def alpha_schedule(time_step, N=100, att_1=0.99999, att_T=0.000009, ctt_1=0.000009, ctt_T=0.9):
    """Mask-and-uniform schedule (:122-151); float64 numpy, as the reference."""
    lin = np.arange(0, time_step) / (time_step - 1)
    att = np.concatenate(([1], lin * (att_T - att_1) + att_1))
    ctt = np.concatenate(([0], lin * (ctt_T - ctt_1) + ctt_1))
    at = att[1:] / att[:-1]
    ct = 1 - (1 - ctt[1:]) / (1 - ctt[:-1])
    bt = (1 - at - ct) / N
    att = np.concatenate((att[1:], [1]))
    ctt = np.concatenate((ctt[1:], [0]))
    btt = (1 - att - ctt) / N
    return at, bt, ct, att, btt, ctt


def _log_1_min_a(a):
    return torch.log(1 - a.exp() + 1e-40)


class DiffusionTransformer(nn.Module):
    def __init__(self, *, content_emb_config=None, condition_emb_config=None, transformer_config=None,
                 diffusion_step=100, alpha_init_type="cos", auxiliary_loss_weight=0,
                 adaptive_auxiliary_loss=False, mask_weight=[1, 1]):
        super().__init__()
        self.condition_emb = instantiate_from_config(condition_emb_config)  # CLIP text: SURVEY 8(f)-1
        transformer_config = dict(transformer_config)
        transformer_config["params"] = dict(transformer_config["params"])
        transformer_config["params"]["diffusion_step"] = diffusion_step
        transformer_config["params"]["content_emb_config"] = content_emb_config
        self.transformer = instantiate_from_config(transformer_config)
        self.content_seq_len = transformer_config["params"]["content_seq_len"]
        self.num_classes = self.transformer.content_emb.num_embed  # K + 1
        self.shape = self.content_seq_len
        self.num_timesteps = diffusion_step
        self.parametrization = "x0"
        self.loss_type = "vb_stochastic"
        self.auxiliary_loss_weight = auxiliary_loss_weight
        self.adaptive_auxiliary_loss = adaptive_auxiliary_loss
        self.mask_weight = mask_weight
        self.truncation_r = None  # set by DALLE.generate_content from sample_type "top{r}r"
        self.truncation_k = None  # ... or "top{k}p" (top-k, dalle_spec.py:147-157); exclusive with truncation_r
        self.repeat_rate = None   # "q{rate}": repeat a step with this probability (dalle_spec.py:135-143)
        # Noise source of the samplers.  "torch" (default): torch.rand((B, K+1, L)) per call on the model's device, the
        # reference's own draw (log_sample_categorical, :359-368) -- same seed, same device => same stream, but what a
        # caption draws depends on the batch around it.  "philox" (or passing caption_ids to sample()): the uniforms are
        # drawn inside the sampler kernel from a counter-based stream keyed by (sample_seed, global caption id, call,
        # position, class) -- a caption's NOISE no longer depends on batch size, batch position or rank (SURVEY.md
        # section 8e), no noise tensor exists, and the whole chain is enqueued by one C call (ds_denoiser_sample_rng).
        # Its tokens are identical across batches as long as the same GEMM program serves the local batch size (the
        # per-sample, half-tile and 4-wave programs agree to ~1e-7 relative on rows 256..264, not bit for bit -- csrc/api.hip
        # rows_per_sample -- so a near-tie can fall differently between, say, one batch of 64 and 8 shards of 8).
        self.rng_mode = "torch"
        self.sample_seed = 1234
        assert alpha_init_type == "alpha1", "Diffsound uses alpha_init_type='alpha1'"
        at, bt, ct, att, btt, ctt = alpha_schedule(self.num_timesteps, N=self.num_classes)
        f64 = lambda x: torch.tensor(x.astype("float64"))
        log_at, log_bt, log_ct = torch.log(f64(at)), torch.log(f64(bt)), torch.log(f64(ct))
        log_cat, log_cbt, log_cct = torch.log(f64(att)), torch.log(f64(btt)), torch.log(f64(ctt))
        for name, v in (("log_at", log_at), ("log_bt", log_bt), ("log_ct", log_ct),
                        ("log_cumprod_at", log_cat), ("log_cumprod_bt", log_cbt), ("log_cumprod_ct", log_cct),
                        ("log_1_min_ct", _log_1_min_a(log_ct)),
                        ("log_1_min_cumprod_ct", _log_1_min_a(log_cct))):
            self.register_buffer(name, v.float())
        self.register_buffer("Lt_history", torch.zeros(self.num_timesteps))
        self.register_buffer("Lt_count", torch.zeros(self.num_timesteps))
        self._sched = None

    @property
    def device(self):
        return self.log_at.device

    def _schedule_table(self):
        """[8][T+1] table consumed by ds_sample_tail (rows: log_at, log_bt, log_ct, log_1_min_ct,
        log_cumprod_at, log_cumprod_bt, log_cumprod_ct, log_1_min_cumprod_ct)."""
        if self._sched is None or self._sched.device != self.log_at.device:
            T = self.num_timesteps
            tab = torch.zeros(8, T + 1, device=self.log_at.device, dtype=torch.float32)
            for i, n in enumerate(("log_at", "log_bt", "log_ct", "log_1_min_ct")):
                tab[i, :T] = getattr(self, n)
            for i, n in enumerate(("log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct", "log_1_min_cumprod_ct")):
                tab[4 + i] = getattr(self, n)
            self._sched = tab
        return self._sched

    # ---- pieces with the reference's signatures (log-one-hot in / out) ------------------------------
    @torch.no_grad()
    def _tail(self, logits_rows, x_t, t, u, initial, want):
        """Run ds_sample_tail on row-major logits; `want` selects debug dumps."""
        B, L, K = x_t.shape[0], self.content_seq_len, self.num_classes - 1
        dev = logits_rows.device
        dump = {k: torch.empty(B, K + 1, L, device=dev) for k in want}
        out = torch.empty(B, L, device=dev, dtype=torch.long)
        r, k = self._truncation()
        _lib.check(_lib.lib().ds_sample_tail_ex(
            _lib.ptr(logits_rows), _lib.ptr(x_t), _lib.ptr(t), _lib.ptr(u), _lib.ptr(self._schedule_table()),
            _lib.ptr(out), _lib.ptr(dump.get("log_pred")), _lib.ptr(dump.get("trunc")), _lib.ptr(dump.get("post")),
            B, L, K, self.num_timesteps, int(initial), r, k, _lib.stream()))
        return out, dump

    def _truncation(self):
        """(trunc_r, trunc_k) for the C ABI: top-k wins if set (the reference installs exactly one wrapper)."""
        if self.truncation_k is not None:
            return -1.0, int(self.truncation_k)
        return (-1.0 if self.truncation_r is None else float(self.truncation_r)), 0

    @staticmethod
    def _is_initial(log_x):
        return bool(torch.isinf(log_x[:, :-1]).all().item())

    @torch.no_grad()
    def step_detail(self, x_t, cond_emb, t, u, initial):
        """Teacher-forced single step on token indices: returns (tokens, {log_pred, trunc, post})."""
        tr = self.transformer
        p = tr.packed(self._schedule_table())
        B = x_t.shape[0]
        x_t, t, u = x_t.contiguous(), t.to(x_t.device).contiguous(), u.contiguous()
        kv = tr.condition_kv(cond_emb, self._schedule_table())
        logits = torch.empty(B * self.content_seq_len, self.num_classes - 1, device=x_t.device)
        _lib.check(_lib.lib().ds_denoiser_forward(p["handle"], _lib.ptr(x_t), _lib.ptr(t), _lib.ptr(kv), B,
                                                  _lib.ptr(tr.workspace(B, self._schedule_table())),
                                                  _lib.ptr(logits), 0, _lib.stream()))
        return self._tail(logits, x_t, t, u, initial, ("log_pred", "trunc", "post"))

    @torch.no_grad()
    def predict_start(self, log_x_t, cond_emb, t):
        """p(x0 | xt) as clamped log-probs [B, K+1, L] (:269-291); with truncation_r set this is the
        reference's truncation-wrapped predict_start (dalle_spec.py:158-174)."""
        x_t = log_x_t.argmax(1)
        u = torch.full((x_t.shape[0], self.num_classes, self.content_seq_len), 0.5, device=x_t.device)
        _, d = self.step_detail(x_t, cond_emb, t, u, self._is_initial(log_x_t))
        return d["trunc"] if (self.truncation_r is not None or self.truncation_k is not None) else d["log_pred"]

    @torch.no_grad()
    def p_sample(self, log_x, cond_emb, t):
        """One reverse step on the reference's log-one-hot state (:353-357)."""
        x_t = log_x.argmax(1)
        u = torch.rand((x_t.shape[0], self.num_classes, self.content_seq_len), device=x_t.device)
        tok = self.p_sample_tokens(x_t, self.transformer.condition_kv(cond_emb, self._schedule_table()), t, u,
                                   self._is_initial(log_x))
        oh = torch.nn.functional.one_hot(tok, self.num_classes).permute(0, 2, 1).float()
        return torch.log(oh.clamp(min=1e-30))

    @torch.no_grad()
    def p_sample_tokens(self, x_t, kv, t, u, initial, out=None, t_post=None, slot=0):
        """x_t i64[B,L] -> x_{t-1} i64[B,L]; kv from transformer.condition_kv().  t_post: the posterior's timestep
        vector when it differs from the network's (sample_fast)."""
        tr = self.transformer
        sched = self._schedule_table()
        p = tr.packed(sched)
        B = x_t.shape[0]
        if out is None:
            out = torch.empty_like(x_t)
        r, k = self._truncation()
        _lib.check(_lib.lib().ds_denoiser_step_ex(p["handle"], _lib.ptr(x_t), _lib.ptr(t), _lib.ptr(t_post), _lib.ptr(kv),
                                                  _lib.ptr(u), B, int(initial), r, k,
                                                  _lib.ptr(tr.workspace(B, sched, slot)), _lib.ptr(out), _lib.stream()))
        return out

    @torch.no_grad()
    def p_sample_tokens_rng(self, x_t, kv, t, caption_ids, call, initial, out=None, t_post=None, slot=0, seed=None):
        """p_sample_tokens with the noise drawn in the kernel: Philox stream of (seed, caption_ids[b], call)."""
        tr = self.transformer
        sched = self._schedule_table()
        p = tr.packed(sched)
        B = x_t.shape[0]
        if out is None:
            out = torch.empty_like(x_t)
        r, k = self._truncation()
        _lib.check(_lib.lib().ds_denoiser_step_rng(
            p["handle"], _lib.ptr(x_t), _lib.ptr(t), _lib.ptr(t_post), _lib.ptr(kv), _lib.ptr(caption_ids),
            int(self.sample_seed if seed is None else seed), int(call), B, int(initial), r, k,
            _lib.ptr(tr.workspace(B, sched, slot)), _lib.ptr(out), _lib.stream()))
        return out

    def _caption_ids(self, caption_ids, B, device):
        """i64[B] global caption ids on the device (default: 0 .. B-1)."""
        if caption_ids is None:
            return torch.arange(B, device=device, dtype=torch.long)
        on_device = torch.is_tensor(caption_ids) and caption_ids.device.type == device.type == "cuda"
        ids = torch.as_tensor(caption_ids, dtype=torch.long)
        if ids.shape != (B,):
            raise ValueError("caption_ids must have one entry per caption: got %s for a batch of %d" % (tuple(ids.shape), B))
        # The range is checked on host lists / CPU tensors every time (it costs nothing there), and on a DEVICE tensor the
        # first time that very tensor OBJECT is seen at its current version (one host synchronisation; remembered through a
        # weak reference to the object, so an id tensor re-used across sample() calls -- bench.py's timed loop -- is checked
        # once, a modified one again, and a NEW tensor that the caching allocator happens to put at the same address is not
        # mistaken for the checked one).  An id outside [0, 2^32) would be truncated into the Philox counter and could share
        # a noise stream with another caption.
        seen = getattr(self, "_gids_checked", None)
        known = on_device and seen is not None and seen[0]() is caption_ids and seen[1] == caption_ids._version
        if B > 0 and not known:
            if int(ids.min()) < 0 or int(ids.max()) >= 2 ** 32:
                raise ValueError("caption ids must be in [0, 2^32)")
            if on_device:
                import weakref
                self._gids_checked = (weakref.ref(caption_ids), caption_ids._version)
        return ids.to(device).contiguous()

    def _cond(self, condition_token, condition_embed):
        if self.condition_emb is not None and condition_token is not None:
            return self.condition_emb(condition_token).float()          # CLIP text tower (:619-621)
        if condition_embed is None:
            raise ValueError("pass condition_token (with a condition_emb module) or condition_embed [B,77,512]")
        return condition_embed.float()

    # ---- training loss: the forward value (the step with gradients is modeling/train.py) ----
    def sample_time(self, b, device, method="uniform", generator=None):
        """Timesteps for a batch and their sampling probabilities (:379-406): importance sampling by sqrt(Lt_history)
        once every timestep has been seen more than 10 times, uniform before.  generator: optional torch.Generator of
        `device` (the reference draws from the global one)."""
        if method == "importance":
            # (Lt_count only grows -- scatter_add of ones --, so once every timestep has been seen 11 times the test stays
            # true: it is read from the device until then, one host synchronisation per call, and never again.  A state-dict
            # load or a manual reset of the statistics clears the memo: _load_from_state_dict / reset_time_statistics.)
            if not getattr(self, "_lt_all_seen", False):
                if not (self.Lt_count > 10).all():
                    return self.sample_time(b, device, method="uniform", generator=generator)
                self._lt_all_seen = True
            lt_sqrt = torch.sqrt(self.Lt_history + 1e-10) + 0.0001
            lt_sqrt[0] = lt_sqrt[1]
            pt_all = lt_sqrt / lt_sqrt.sum()
            t = torch.multinomial(pt_all, num_samples=b, replacement=True, generator=generator)
            return t, pt_all.gather(dim=0, index=t)
        if method == "uniform":
            t = torch.randint(0, self.num_timesteps, (b,), device=device, generator=generator).long()
            return t, torch.ones_like(t).float() / self.num_timesteps
        raise ValueError(method)

    def reset_time_statistics(self):
        """Zero the importance-sampling statistics (:88-89) and forget that every timestep had been seen."""
        self.Lt_history.zero_()
        self.Lt_count.zero_()
        self._lt_all_seen = False

    def _load_from_state_dict(self, *args, **kwargs):
        self._lt_all_seen = False
        return super()._load_from_state_dict(*args, **kwargs)

    @torch.no_grad()
    def _train_loss(self, x, cond_emb, is_train=True, noise=None):
        """(log_model_prob [B, K+1, L], vb_loss [B]) of :408-476, computed forward-only on the HIP path: ds_q_sample,
        the denoiser, and ds_loss_tail for the per-position KL / NLL / auxiliary-KL terms.  Updates Lt_history /
        Lt_count like the reference (the accuracy bookkeeping lists :425-436 are logging only and not kept).
        noise: optional f32[B, K+1, L] uniforms for q_sample (tests)."""
        B, L, K1, T = x.shape[0], self.content_seq_len, self.num_classes, self.num_timesteps
        dev = x.device
        t, pt = self.sample_time(B, dev, "importance")
        t, pt = t.to(dev), pt.to(dev)
        u = torch.rand((B, K1, L), device=dev) if noise is None else noise.to(dev)
        x = x.contiguous()
        xt = self.q_sample_tokens(x, t, u)
        tr = self.transformer
        sched = self._schedule_table()
        p = tr.packed(sched)
        kv = tr.condition_kv(cond_emb, sched)
        logits = torch.empty(B * L, K1 - 1, device=dev)
        _lib.check(_lib.lib().ds_denoiser_forward(p["handle"], _lib.ptr(xt), _lib.ptr(t), _lib.ptr(kv), B,
                                                  _lib.ptr(tr.workspace(B, sched)), _lib.ptr(logits), 0, _lib.stream()))
        kl, nll, kl_aux = (torch.empty(B, L, device=dev) for _ in range(3))
        log_model_prob = torch.empty(B, K1, L, device=dev)
        _lib.check(_lib.lib().ds_loss_tail(_lib.ptr(logits), _lib.ptr(x), _lib.ptr(xt), _lib.ptr(t),
                                           _lib.ptr(sched), _lib.ptr(kl), _lib.ptr(nll), _lib.ptr(kl_aux),
                                           _lib.ptr(log_model_prob), B, L, K1 - 1, T, _lib.stream()))
        mask_region = (xt == K1 - 1).float()
        weight = mask_region * self.mask_weight[0] + (1.0 - mask_region) * self.mask_weight[1]
        is0 = (t == 0).float()
        decoder_nll = nll.sum(-1)
        kl_loss = is0 * decoder_nll + (1.0 - is0) * (kl * weight).sum(-1)
        lt2 = kl_loss.pow(2)
        self.Lt_history.scatter_(dim=0, index=t, src=(0.1 * lt2 + 0.9 * self.Lt_history.gather(dim=0, index=t)))
        self.Lt_count.scatter_add_(dim=0, index=t, src=torch.ones_like(lt2))
        vb_loss = kl_loss / pt
        if self.auxiliary_loss_weight != 0 and is_train:
            kl_aux_loss = is0 * decoder_nll + (1.0 - is0) * (kl_aux * weight).sum(-1)
            w = t.float() / T + 1.0 if self.adaptive_auxiliary_loss else 1.0
            vb_loss = vb_loss + w * self.auxiliary_loss_weight * kl_aux_loss / pt
        return log_model_prob, vb_loss

    @torch.no_grad()
    def forward(self, input, return_loss=False, return_logits=True, return_att_weight=False, is_train=True, **kwargs):
        """{'logits': exp(log_model_prob), 'loss': scalar} as :539-577.  The loss is a forward value: this package
        has no backward pass, so it serves evaluation / loss parity, not optimisation."""
        x = input["content_token"]
        cond_emb = self._cond(input.get("condition_token"), input.get("condition_embed_token")).to(x.device)
        out = {}
        if is_train:
            log_model_prob, loss = self._train_loss(x, cond_emb, noise=kwargs.get("noise"))
            loss = loss.sum() / (x.shape[0] * x.shape[1])
            if return_logits:
                out["logits"] = torch.exp(log_model_prob)
            if return_loss:
                out["loss"] = loss
        return out

    @torch.no_grad()
    def q_sample_tokens(self, x0, t, u):
        """x_t ~ q(x_t | x_0) on token ids (q_sample, :370-377); u f32[B, K+1, L] uniforms."""
        x0_c, t_c, u_c = x0.contiguous(), t.contiguous(), u.contiguous()   # named: alive until the launch is enqueued
        out = torch.empty_like(x0_c)
        _lib.check(_lib.lib().ds_q_sample(_lib.ptr(x0_c), _lib.ptr(t_c), _lib.ptr(u_c),
                                          _lib.ptr(self._schedule_table()), _lib.ptr(out), x0_c.shape[0],
                                          self.content_seq_len, self.num_classes - 1, self.num_timesteps,
                                          _lib.stream()))
        return out

    @torch.no_grad()
    def q_sample(self, log_x_start, t):
        """The reference's log-one-hot form of the forward diffusion (:370-377): log_x_start f32[B, K+1, L] one-hot in
        log space -> a log-one-hot sample of q(x_t | x_0), drawing torch.rand of the same shape as the reference."""
        x0 = log_x_start.argmax(1)
        u = torch.rand(tuple(log_x_start.shape), device=x0.device)
        xt = self.q_sample_tokens(x0, t.to(x0.device), u)
        oh = torch.nn.functional.one_hot(xt, self.num_classes).permute(0, 2, 1).float()
        return torch.log(oh.clamp(min=1e-30))

    def _reverse(self, cond_emb, steps, noise_fn, return_logits, start_tokens=None, caption_ids=None, seed=None):
        """steps: list of (t, t_post) pairs, first one from the all-[MASK] state (or from start_tokens, already
        diffused to the first t).  The 'q' repeat sampler
        (dalle_spec.py:135-143: with probability `repeat_rate` a step is applied twice at the same t) draws from
        Python's `random` exactly like the reference's wrapper: one random.random() per step."""
        import random
        device = self.device
        B = cond_emb.shape[0]
        K1, L = self.num_classes, self.content_seq_len
        cond_emb = cond_emb.to(device)
        if start_tokens is None:
            x = torch.full((B, L), K1 - 1, device=device, dtype=torch.long)  # all [MASK]
        else:
            x = start_tokens.to(device).clone()
        if noise_fn is None and (caption_ids is not None or self.rng_mode == "philox"):
            # in-kernel noise: the whole chain is one C call, nothing returns to Python between steps
            calls = []
            for step, step_post in steps:
                reps = 2 if (self.repeat_rate is not None and random.random() < self.repeat_rate) else 1
                calls += [(step, step_post)] * reps
            sched = self._schedule_table()
            tr = self.transformer
            p = tr.packed(sched)
            kv = tr.condition_kv(cond_emb.contiguous(), sched)
            gids = self._caption_ids(caption_ids, B, device)
            t_steps = torch.tensor(calls, dtype=torch.long, device=device).view(-1, 2, 1).expand(-1, 2, B).contiguous()
            tmp = torch.empty_like(x)
            r, k = self._truncation()
            _lib.check(_lib.lib().ds_denoiser_sample_rng(
                p["handle"], _lib.ptr(x), _lib.ptr(tmp), _lib.ptr(t_steps), len(calls), _lib.ptr(kv), _lib.ptr(gids),
                int(self.sample_seed if seed is None else seed), 0, B, int(start_tokens is None), r, k,
                _lib.ptr(tr.workspace(B, sched, 0)), _lib.stream()))
            out = {"content_token": x}
            if return_logits:
                out["logits"] = torch.nn.functional.one_hot(x, K1).permute(0, 2, 1).float()
            return out
        sched = self._schedule_table()
        kv = self.transformer.condition_kv(cond_emb.contiguous(), sched)
        nxt = torch.empty_like(x)
        calls = 0
        for i, (step, step_post) in enumerate(steps):
            repeats = 2 if (self.repeat_rate is not None and random.random() < self.repeat_rate) else 1
            for rep in range(repeats):
                # the reference's draw: torch.rand_like(logits) on a [B, K+1, L] fp32 tensor of the model's device
                # (diffusion_transformer.py:359-368) -- torch.rand of that shape consumes the generator identically
                u = noise_fn(step if self.repeat_rate is None else calls, (B, K1, L)).to(device) \
                    if noise_fn is not None else torch.rand((B, K1, L), device=device)
                u = u.contiguous()
                calls += 1
                t = torch.full((B,), step, device=device, dtype=torch.long)
                tp = None if step_post == step else torch.full((B,), step_post, device=device, dtype=torch.long)
                self.p_sample_tokens(x, kv, t, u, initial=(start_tokens is None and i == 0 and rep == 0), out=nxt, t_post=tp)
                x, nxt = nxt, x
        out = {"content_token": x}
        if return_logits:
            out["logits"] = torch.nn.functional.one_hot(x, K1).permute(0, 2, 1).float()
        return out

    @torch.no_grad()
    def sample(self, condition_token, condition_mask, condition_embed, content_token=None, filter_ratio=0.5,
               temperature=1.0, return_att_weight=False, return_logits=False, content_logits=None,
               print_log=True, noise_fn=None, caption_ids=None, seed=None, **kwargs):
        """Reverse diffusion from the all-[MASK] state (:587-659, filter_ratio == 0 branch).

        noise_fn(step, shape) may supply the uniforms (tests inject the oracle's noise; with the 'q' repeat sampler
        active the first argument is the running p_sample call index instead of the timestep).  caption_ids (i64[B],
        global caption indices) selects the in-kernel per-caption noise (see rng_mode); seed overrides sample_seed."""
        cond_emb = self._cond(condition_token, condition_embed)
        T = self.num_timesteps
        start_step = int(T * filter_ratio)
        if start_step == 0:
            return self._reverse(cond_emb, [(s_, s_) for s_ in range(T - 1, -1, -1)], noise_fn, return_logits,
                                 caption_ids=caption_ids, seed=seed)
        # partial re-sampling (:643-651): diffuse the given tokens (e.g. DALLE.get_tokens of a mel) forward to
        # t = start_step - 1, then run the reverse chain from there.  With noise_fn, call 0 is q_sample's draw and
        # the reverse steps get the running call index 1.. instead of the timestep.
        if content_token is None:
            raise ValueError("filter_ratio > 0 re-samples given content: pass content_token i64[B, 265]")
        device, B = self.device, cond_emb.shape[0]
        shape = (B, self.num_classes, self.content_seq_len)
        t = torch.full((B,), start_step - 1, device=device, dtype=torch.long)
        steps = list(range(start_step - 1, -1, -1))
        if noise_fn is None and (caption_ids is not None or self.rng_mode == "philox"):
            # forward diffusion on stream 1 of each caption's Philox draws (call 0), the reverse chain on stream 0
            x0 = content_token.to(device).contiguous()
            gids = self._caption_ids(caption_ids, B, device)
            x = torch.empty_like(x0)
            _lib.check(_lib.lib().ds_q_sample_rng(_lib.ptr(x0), _lib.ptr(t), _lib.ptr(gids),
                                                  int(self.sample_seed if seed is None else seed), 0,
                                                  _lib.ptr(self._schedule_table()), _lib.ptr(x), B, self.content_seq_len,
                                                  self.num_classes - 1, self.num_timesteps, _lib.stream()))
            return self._reverse(cond_emb, [(s_, s_) for s_ in steps], None, return_logits, start_tokens=x,
                                 caption_ids=gids, seed=seed)
        u = noise_fn(0, shape).to(device) if noise_fn is not None else torch.rand(shape, device=device)
        x = self.q_sample_tokens(content_token.to(device), t, u)
        nf = None if noise_fn is None else (lambda st, shp: noise_fn(1 + steps.index(st), shp)) \
            if self.repeat_rate is None else (lambda c, shp: noise_fn(1 + c, shp))
        return self._reverse(cond_emb, [(s_, s_) for s_ in steps], nf, return_logits, start_tokens=x)

    @torch.no_grad()
    def sample_fast(self, condition_token, condition_mask, condition_embed, content_token=None, filter_ratio=0.5,
                    temperature=1.0, return_att_weight=False, return_logits=False, content_logits=None,
                    print_log=True, skip_step=1, noise_fn=None, caption_ids=None, seed=None, **kwargs):
        """Skip-step sampler (:748-812): timesteps T-1, T-2-skip, ... (0 appended), the network sees t, the
        posterior t - skip_step while t > skip_step.  The reference calls p_pred's pieces directly, so the 'q'
        wrapper on p_sample never applies here."""
        cond_emb = self._cond(condition_token, condition_embed)
        assert int(self.num_timesteps * filter_ratio) == 0     # the reference asserts start_step == 0 (:787)
        lst = list(range(self.num_timesteps - 1, -1, -1 - skip_step))
        if lst[-1] != 0:
            lst.append(0)
        keep, self.repeat_rate = self.repeat_rate, None
        try:
            return self._reverse(cond_emb, [(s_, s_ - skip_step if s_ > skip_step else s_) for s_ in lst], noise_fn,
                                 return_logits, caption_ids=caption_ids, seed=seed)
        finally:
            self.repeat_rate = keep
