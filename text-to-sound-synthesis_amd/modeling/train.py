"""Training step of the denoiser on the HIP kernels (scope row 8f-3).  The linear layers -- forward, dX = dY W and
dW = dY^T X, 98 % of the step's flops -- run on the exact-fp32 MFMA GEMM (default) or, with
`TrainStep(precision="f16x2")`, on the fp32-class 3-pass fp16 split GEMM (`ds_gemm_f16x2`, loader-split A with the
gradients rescaled by an exact power of two into fp16's range, fp32 accumulate).  Measured on MI355X at the
reference's batch 20 (tools/bench_train.py, profiles/r02_bench_train_*.json): 4.06 it/s fp32, 3.51 it/s f16x2 -- the
split GEMMs themselves are ~2x faster, but this host-composed step re-splits weights and transposed activations with
torch ops and a host sync per GEMM, and at M = 5300 rows that overhead outweighs the gain.  Worst per-tensor gradient
error against autograd through the oracle: 4.7e-6 (fp32), 6.9e-5 (f16x2).  Attention, norms and the loss tail are exact
fp32 in both modes.

    loss, grads = TrainStep(model.transformer).loss_and_grads(x0, cond_emb, t, pt, noise)

follows DiffusionTransformer._train_loss / forward (diffusion_transformer.py:408-476,539-577) and what
`loss.backward()` produces for every parameter of `Text2ImageTransformer` (engine/solver_spec.py back-propagates
exactly this).  Every GEMM-shaped contraction of the forward and the backward runs on the fp32-MFMA gather-GEMM
(`ds_gemm`), the row / elementwise pieces on csrc/train.hip, norm.hip and sampler.hip; torch is used for memory and
for layout copies (transposes, head split / merge, zero padding to the GEMM's 32-wide K granule) and for the
100-row timestep-embedding table, which is weight preparation.  Nothing here is tuned: the packed-plane / LDS-DMA
machinery of the sampling path is not used, activations are kept in fp32, attention probabilities are materialised.
"""
import torch

from .. import _lib

L_ = _lib


_SPLIT = [True]      # set by TrainStep: linear-layer GEMMs on the 3-pass fp16 split (default) or on the fp32 MFMA


def _pow2_into_fp16_range(a):
    """(a * 2^k, 2^-k) with k an integer that puts max|a| into [2^9, 2^10) when it is outside [2^-6, 2^13) -- the
    split GEMM's A operand is decomposed into two fp16 planes and needs its values inside fp16's range with headroom;
    gradients (|dY| ~ 1e-8 .. 1e-3) are far below it.  Multiplying by a power of two is exact."""
    import math
    mx = float(a.abs().max())
    if mx == 0.0 or not math.isfinite(mx) or 2.0 ** -6 <= mx < 2.0 ** 13:
        return a, 1.0
    k = 9 - math.floor(math.log2(mx))
    return a * (2.0 ** k), 2.0 ** (-k)


def _mm_nt(A, Wm, out, M, N, K, bias=None, R=None, act=L_.ACT_NONE):
    """out[M, N] = act(A[M, K] Wm[N, K]^T + bias) + R"""
    if not _SPLIT[0]:
        return L_.gemm(A, Wm, out, M, N, K, bias=bias, R=R, act=act)
    W2, sc = L_.split_f16x2(Wm)
    A2, sa = _pow2_into_fp16_range(A)
    return L_.gemm(A2, W2, out, M, N, K, bias=bias, R=R, act=act, split2=sc * sa)


def _lin_fwd(x, W, b, R=None, act=L_.ACT_NONE):
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, device=x.device)
    _mm_nt(x, W, y, M, N, K, bias=b, R=R, act=act)
    return y


def _pad_rows_t(a, kp):
    """a [M, C] -> a^T zero-padded to [C, kp] (kp = M rounded up to the GEMM's K granule)"""
    out = torch.zeros(a.shape[1], kp, device=a.device)
    out[:, :a.shape[0]] = a.t()
    return out


def _colsum(x, G=1, R=None, accumulate_into=None):
    M, C_ = x.shape
    R = M // G if R is None else R
    out = torch.empty(G, C_, device=x.device) if accumulate_into is None else accumulate_into
    L_.check(L_.lib().ds_colsum(L_.ptr(x), L_.ptr(out), G, R, C_, C_, R * C_, int(accumulate_into is not None), L_.stream()))
    return out


def _lin_bwd(x, W, dy, need_dx=True):
    """y = x W^T + b  ->  (dx = dy W, dW = dy^T x, db = column sums of dy)"""
    M, K = x.shape
    N = W.shape[0]
    dx = None
    if need_dx:
        dx = torch.empty(M, K, device=x.device)
        Wt = W.t().contiguous()                                   # [K, N]: rows are K-contiguous operands of the GEMM
        Np = (N + 31) // 32 * 32
        if Np != N:
            Wt = torch.nn.functional.pad(Wt, (0, Np - N))
            dyp = torch.nn.functional.pad(dy, (0, Np - N))
        else:
            dyp = dy
        _mm_nt(dyp.contiguous(), Wt.contiguous(), dx, M, K, Np)
    Mp = (M + 31) // 32 * 32
    dW = torch.empty(N, K, device=x.device)
    _mm_nt(_pad_rows_t(dy, Mp), _pad_rows_t(x, Mp), dW, N, K, Mp)
    return dx, dW, _colsum(dy)[0]


def _norm_fwd(x, mode, L, table=None, t=None, gamma=None, beta=None):
    M, D = x.shape
    y = torch.empty_like(x)
    if mode == 0:
        L_.check(L_.lib().ds_adaln(L_.ptr(x), L_.ptr(y), M, L, D, L_.ptr(table), L_.ptr(t), L_.stream()))
    else:
        L_.check(L_.lib().ds_layernorm(L_.ptr(x), L_.ptr(y), M, D, L_.ptr(gamma), L_.ptr(beta), L_.stream()))
    return y


def _norm_bwd(x, dy, mode, L, table=None, t=None, gamma=None):
    M, D = x.shape
    dx, dyxn = torch.empty_like(x), torch.empty_like(x)
    L_.check(L_.lib().ds_layernorm_bwd(L_.ptr(x), L_.ptr(dy), L_.ptr(dx), L_.ptr(dyxn), M, L, D, mode, L_.ptr(table),
                                       L_.ptr(t), L_.ptr(gamma), L_.stream()))
    G = M // L if mode == 0 else 1
    return dx, _colsum(dyxn, G), _colsum(dy, G)               # d scale, d shift per sample (AdaLN) or summed (LN)


def _heads(x, B, Lx, H, Lp):
    """[B*Lx, H*64] -> [B*H, Lp, 64] zero-padded"""
    out = torch.zeros(B * H, Lp, 64, device=x.device)
    out[:, :Lx] = x.view(B, Lx, H, 64).permute(0, 2, 1, 3).reshape(B * H, Lx, 64)
    return out


def _merge(x4, B, Lx, H):
    return x4[:, :Lx].reshape(B, H, Lx, 64).permute(0, 2, 1, 3).reshape(B * Lx, H * 64).contiguous()


class _Attn:
    """softmax(q k^T / 8) v per head (FullAttention / CrossAttention cores, transformer_utils.py:43-58,91-109)"""

    def __init__(self, q, k, v, B, Lq, Lk, H):
        self.B, self.Lq, self.Lk, self.H = B, Lq, Lk, H
        self.Lqp, self.Lkp = (Lq + 31) // 32 * 32, (Lk + 31) // 32 * 32
        G = B * H
        self.q4, self.k4, self.v4 = _heads(q, B, Lq, H, self.Lqp), _heads(k, B, Lk, H, self.Lkp), _heads(v, B, Lk, H, self.Lkp)
        S = torch.empty(G, self.Lqp, self.Lkp, device=q.device)
        L_.gemm(self.q4, self.k4, S, self.Lqp, self.Lkp, 64, groups=G, a_gstride=self.Lqp * 64, w_gstride=self.Lkp * 64,
                c_gstride=self.Lqp * self.Lkp)
        L_.check(L_.lib().ds_softmax_rows(L_.ptr(S), G * self.Lqp, Lk, self.Lkp, 0.125, L_.stream()))
        self.P = S
        vT = self.v4.transpose(1, 2).contiguous()                                  # [G, 64, Lkp]
        o4 = torch.empty(G, self.Lqp, 64, device=q.device)
        L_.gemm(self.P, vT, o4, self.Lqp, 64, self.Lkp, groups=G, a_gstride=self.Lqp * self.Lkp, w_gstride=64 * self.Lkp,
                c_gstride=self.Lqp * 64)
        self.out = _merge(o4, B, Lq, H)

    def backward(self, dO):
        B, Lq, Lk, H, Lqp, Lkp = self.B, self.Lq, self.Lk, self.H, self.Lqp, self.Lkp
        G = B * H
        dO4 = _heads(dO, B, Lq, H, Lqp)
        dev = dO.device
        dV4 = torch.empty(G, Lkp, 64, device=dev)                                   # dV = P^T dO
        L_.gemm(self.P.transpose(1, 2).contiguous(), dO4.transpose(1, 2).contiguous(), dV4, Lkp, 64, Lqp, groups=G,
                a_gstride=Lkp * Lqp, w_gstride=64 * Lqp, c_gstride=Lkp * 64)
        dP = torch.empty(G, Lqp, Lkp, device=dev)                                   # dP = dO V^T
        L_.gemm(dO4, self.v4, dP, Lqp, Lkp, 64, groups=G, a_gstride=Lqp * 64, w_gstride=Lkp * 64, c_gstride=Lqp * Lkp)
        L_.check(L_.lib().ds_softmax_bwd_rows(L_.ptr(self.P), L_.ptr(dP), G * Lqp, Lk, Lkp, 0.125, L_.stream()))
        dS = dP
        dQ4 = torch.empty(G, Lqp, 64, device=dev)                                   # dQ = dS K
        L_.gemm(dS, self.k4.transpose(1, 2).contiguous(), dQ4, Lqp, 64, Lkp, groups=G, a_gstride=Lqp * Lkp,
                w_gstride=64 * Lkp, c_gstride=Lqp * 64)
        dK4 = torch.empty(G, Lkp, 64, device=dev)                                   # dK = dS^T Q
        L_.gemm(dS.transpose(1, 2).contiguous(), self.q4.transpose(1, 2).contiguous(), dK4, Lkp, 64, Lqp, groups=G,
                a_gstride=Lkp * Lqp, w_gstride=64 * Lqp, c_gstride=Lkp * 64)
        return _merge(dQ4, B, Lq, H), _merge(dK4, B, Lk, H), _merge(dV4, B, Lk, H)


class TrainStep:
    def __init__(self, diffusion_transformer, precision="fp32"):
        assert precision in ("f16x2", "fp32")
        self.dt = diffusion_transformer
        self.tr = diffusion_transformer.transformer
        self.precision = precision

    @torch.no_grad()
    def loss_and_grads(self, x0, cond_emb, t, pt, noise):
        """x0 i64[B, L] clean tokens, cond_emb f32[B, 77, 512], t i64[B], pt f32[B] (sample_time's output), noise
        f32[B, K+1, L] uniforms for q_sample.  Returns (loss scalar as forward() reports it, {parameter name relative to
        the DiffusionTransformer: gradient}).  Gradients are those of that loss."""
        dt, tr = self.dt, self.tr
        _SPLIT[0] = self.precision == "f16x2"
        dev = x0.device
        B, Lx = x0.shape
        D, H, K = tr.n_embd, tr.n_head, tr.num_codes
        M = B * Lx
        T = dt.num_timesteps
        sched = dt._schedule_table()
        xt = dt.q_sample_tokens(x0.contiguous(), t, noise)
        emb = tr.content_emb
        pos = emb.position_table()
        x = torch.empty(M, D, device=dev)
        L_.check(L_.lib().ds_embed(L_.ptr(xt), L_.ptr(emb.emb.weight), L_.ptr(pos), L_.ptr(x), M, Lx, D, L_.stream()))
        cond = cond_emb.reshape(-1, cond_emb.shape[-1]).float().contiguous()
        Lc = cond_emb.shape[1]
        saved = []
        for blk in tr.blocks:
            s = {"x0": x}
            s["tab1"] = blk.ln1.table()
            h = _norm_fwd(x, 0, Lx, table=s["tab1"], t=t)
            a1 = blk.attn1
            s["h1"] = h
            q, k, v = _lin_fwd(h, a1.query.weight, a1.query.bias), _lin_fwd(h, a1.key.weight, a1.key.bias), \
                _lin_fwd(h, a1.value.weight, a1.value.bias)
            s["att1"] = _Attn(q, k, v, B, Lx, Lx, H)
            x = _lin_fwd(s["att1"].out, a1.proj.weight, a1.proj.bias, R=x)
            s["x1"] = x
            s["tab2"] = blk.ln1_1.table()
            h = _norm_fwd(x, 0, Lx, table=s["tab2"], t=t)
            s["h2"] = h
            a2 = blk.attn2
            q = _lin_fwd(h, a2.query.weight, a2.query.bias)
            k, v = _lin_fwd(cond, a2.key.weight, a2.key.bias), _lin_fwd(cond, a2.value.weight, a2.value.bias)
            s["att2"] = _Attn(q, k, v, B, Lx, Lc, H)
            x = _lin_fwd(s["att2"].out, a2.proj.weight, a2.proj.bias, R=x)
            s["x2"] = x
            h = _norm_fwd(x, 1, Lx, gamma=blk.ln2.weight, beta=blk.ln2.bias)
            s["h3"] = h
            u = _lin_fwd(h, blk.mlp[0].weight, blk.mlp[0].bias)
            s["u"] = u
            gact = torch.empty_like(u)
            L_.check(L_.lib().ds_gelu2(L_.ptr(u), None, L_.ptr(gact), u.numel(), L_.stream()))
            s["g"] = gact
            x = _lin_fwd(gact, blk.mlp[2].weight, blk.mlp[2].bias, R=x)
            saved.append(s)
        xf = x
        lnf, lin = tr.to_logits[0], tr.to_logits[1]
        hf = _norm_fwd(xf, 1, Lx, gamma=lnf.weight, beta=lnf.bias)
        logits = _lin_fwd(hf, lin.weight, lin.bias)                                  # [M, K]
        # ---- loss (forward value) and d loss / d logits
        kl, nll, kl_aux = (torch.empty(B, Lx, device=dev) for _ in range(3))
        L_.check(L_.lib().ds_loss_tail(L_.ptr(logits), L_.ptr(x0), L_.ptr(xt), L_.ptr(t), L_.ptr(sched), L_.ptr(kl), L_.ptr(nll),
                                       L_.ptr(kl_aux), None, B, Lx, K, T, L_.stream()))
        mask_region = (xt == K).float()
        weight = mask_region * dt.mask_weight[0] + (1.0 - mask_region) * dt.mask_weight[1]
        is0 = (t == 0).float()
        kl_loss = is0 * nll.sum(-1) + (1.0 - is0) * (kl * weight).sum(-1)
        lt2 = kl_loss.pow(2)                         # importance-sampling statistics of sample_time (:452-455)
        dt.Lt_history.scatter_(dim=0, index=t, src=(0.1 * lt2 + 0.9 * dt.Lt_history.gather(dim=0, index=t)))
        dt.Lt_count.scatter_add_(dim=0, index=t, src=torch.ones_like(lt2))
        vb = kl_loss / pt
        if dt.auxiliary_loss_weight != 0:
            wa = t.float() / T + 1.0 if dt.adaptive_auxiliary_loss else 1.0
            vb = vb + wa * dt.auxiliary_loss_weight * (is0 * nll.sum(-1) + (1.0 - is0) * (kl_aux * weight).sum(-1)) / pt
        norm = 1.0 / (B * Lx)
        loss = vb.sum() * norm
        dlog = torch.empty(M, K, device=dev)
        L_.check(L_.lib().ds_loss_tail_bwd(L_.ptr(logits), L_.ptr(x0), L_.ptr(xt), L_.ptr(t), L_.ptr(pt.contiguous()),
                                           L_.ptr(sched), L_.ptr(dlog), B, Lx, K, T, float(dt.mask_weight[0]),
                                           float(dt.mask_weight[1]), float(dt.auxiliary_loss_weight),
                                           int(bool(dt.adaptive_auxiliary_loss)), L_.stream()))
        dlog.mul_(norm)                                                              # loss = sum(vb) / (B L)
        # ---- backward
        g = {}
        dh, g["transformer.to_logits.1.weight"], g["transformer.to_logits.1.bias"] = _lin_bwd(hf, lin.weight, dlog)
        dx, g["transformer.to_logits.0.weight"], g["transformer.to_logits.0.bias"] = \
            (lambda r: (r[0], r[1][0], r[2][0]))(_norm_bwd(xf, dh, 1, Lx, gamma=lnf.weight))

        def axpy(y, x_):
            L_.check(L_.lib().ds_axpy(L_.ptr(y), L_.ptr(x_), 1.0, y.numel(), L_.stream()))

        def adaln_param_grads(ln, table_rows_scale, table_rows_shift, pfx):
            """d table[t_b] rows -> emb.weight / linear.{weight, bias} through table = Linear(SiLU(emb)) (weight prep)."""
            dtab = torch.zeros(T, 2 * D, device=dev)
            dtab.index_add_(0, t, torch.cat((table_rows_scale, table_rows_shift), dim=1))
            with torch.enable_grad():
                e = ln.emb.weight.detach().clone().requires_grad_(True)
                w = ln.linear.weight.detach().clone().requires_grad_(True)
                b = ln.linear.bias.detach().clone().requires_grad_(True)
                tab = torch.nn.functional.linear(torch.nn.functional.silu(e), w, b)
                tab.backward(dtab)
            g[pfx + ".emb.weight"], g[pfx + ".linear.weight"], g[pfx + ".linear.bias"] = e.grad, w.grad, b.grad

        for li in reversed(range(len(saved))):
            s, blk = saved[li], tr.blocks[li]
            p = "transformer.blocks.%d." % li
            # x3 = x2 + fc2(gelu(fc1(ln2(x2))))
            dgact, g[p + "mlp.2.weight"], g[p + "mlp.2.bias"] = _lin_bwd(s["g"], blk.mlp[2].weight, dx)
            du = torch.empty_like(dgact)
            L_.check(L_.lib().ds_gelu2(L_.ptr(s["u"]), L_.ptr(dgact), L_.ptr(du), du.numel(), L_.stream()))
            dh, g[p + "mlp.0.weight"], g[p + "mlp.0.bias"] = _lin_bwd(s["h3"], blk.mlp[0].weight, du)
            dxn, dgam, dbet = _norm_bwd(s["x2"], dh, 1, Lx, gamma=blk.ln2.weight)
            g[p + "ln2.weight"], g[p + "ln2.bias"] = dgam[0], dbet[0]
            axpy(dx, dxn)
            # x2 = x1 + proj2(attn2(q(ln1_1(x1)), kv(cond)))
            a2 = blk.attn2
            dao, g[p + "attn2.proj.weight"], g[p + "attn2.proj.bias"] = _lin_bwd(s["att2"].out, a2.proj.weight, dx)
            dq, dk, dv = s["att2"].backward(dao)
            dh, g[p + "attn2.query.weight"], g[p + "attn2.query.bias"] = _lin_bwd(s["h2"], a2.query.weight, dq)
            _, g[p + "attn2.key.weight"], g[p + "attn2.key.bias"] = _lin_bwd(cond, a2.key.weight, dk, need_dx=False)
            _, g[p + "attn2.value.weight"], g[p + "attn2.value.bias"] = _lin_bwd(cond, a2.value.weight, dv, need_dx=False)
            dxn, dsc, dsh = _norm_bwd(s["x1"], dh, 0, Lx, table=s["tab2"], t=t)
            adaln_param_grads(blk.ln1_1, dsc, dsh, p + "ln1_1")
            axpy(dx, dxn)
            # x1 = x0 + proj1(attn1(qkv(ln1(x0))))
            a1 = blk.attn1
            dao, g[p + "attn1.proj.weight"], g[p + "attn1.proj.bias"] = _lin_bwd(s["att1"].out, a1.proj.weight, dx)
            dq, dk, dv = s["att1"].backward(dao)
            dh, g[p + "attn1.query.weight"], g[p + "attn1.query.bias"] = _lin_bwd(s["h1"], a1.query.weight, dq)
            dh2, g[p + "attn1.key.weight"], g[p + "attn1.key.bias"] = _lin_bwd(s["h1"], a1.key.weight, dk)
            dh3, g[p + "attn1.value.weight"], g[p + "attn1.value.bias"] = _lin_bwd(s["h1"], a1.value.weight, dv)
            axpy(dh, dh2)
            axpy(dh, dh3)
            dxn, dsc, dsh = _norm_bwd(s["x0"], dh, 0, Lx, table=s["tab1"], t=t)
            adaln_param_grads(blk.ln1, dsc, dsh, p + "ln1")
            axpy(dx, dxn)
        # ---- embedding
        demb = torch.zeros_like(emb.emb.weight)
        L_.check(L_.lib().ds_embed_bwd(L_.ptr(dx), L_.ptr(xt), L_.ptr(demb), M, D, demb.shape[0], L_.stream()))
        g["transformer.content_emb.emb.weight"] = demb
        dpos = torch.empty(Lx, D, device=dev)
        L_.check(L_.lib().ds_colsum(L_.ptr(dx), L_.ptr(dpos), Lx, B, D, Lx * D, D, 0, L_.stream()))
        Hh, Ww = emb.spatial_size
        dpos3 = dpos.view(Hh, Ww, D)
        g["transformer.content_emb.height_emb.weight"] = _colsum(dpos3.reshape(Hh * Ww, D), Hh)       # sum over w
        dw = torch.empty(Ww, D, device=dev)
        L_.check(L_.lib().ds_colsum(L_.ptr(dpos), L_.ptr(dw), Ww, Hh, D, Ww * D, D, 0, L_.stream()))   # sum over h
        g["transformer.content_emb.width_emb.weight"] = dw
        return loss, g

    @torch.no_grad()
    def adamw_step(self, grads, state, step, lr, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2):
        """In-place AdamW on the parameters that have a gradient (state: dict name -> (m, v), created on first use)."""
        params = dict(self.dt.named_parameters())
        for name, gr in grads.items():
            p_ = params[name]
            if name not in state:
                state[name] = (torch.zeros_like(p_), torch.zeros_like(p_))
            m, v = state[name]
            gr = gr.contiguous()
            L_.check(L_.lib().ds_adamw(L_.ptr(p_.data), L_.ptr(gr), L_.ptr(m), L_.ptr(v), p_.numel(), lr, betas[0], betas[1],
                                       eps, weight_decay, step, L_.stream()))
        self.tr.invalidate()       # cached weight packs / AdaLN tables are stale now (frees the native handle too)
