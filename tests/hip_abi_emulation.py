"""TEST INFRASTRUCTURE: a torch-CPU emulation of the C-ABI entries that the training step (modeling/train.py) calls, so
that its host composition -- operand layouts, fused projections, split-K launches, the loss scale, the gradient dict --
runs in the CPU suite, where there is no GPU.  It is NOT a fallback of the product: the package has no CPU path, this
module lives under tests/, and it is installed by monkeypatching `text_to_sound_synthesis_amd._lib` inside a test.

Every function follows the contract written in include/diffsound_hip.h for its entry (argument order of `_lib._PROTOS`);
pointers are the tensors themselves (`ptr = identity`), addressed as flat storage with the strides the kernels get.  The
loss tail comes from the oracle (diffsound_oracle.train_loss's pieces / loss_tail_backward).
"""
import math

import torch

import diffsound_oracle as O


def _flat(t):
    assert t.is_contiguous()
    return t.reshape(-1)


def _split16(a):
    """the f16x2 split of an fp32 tensor, as the GEMM's loader / ds_convert_operand compute it (common.h ds_split_*)"""
    hi = a.clamp(-65504.0, 65504.0).half()
    lo = (a - hi.float()).clamp(-65504.0, 65504.0).half()
    return hi, lo


def _sched_dict(tab, T):
    names = ("log_at", "log_bt", "log_ct", "log_1_min_ct", "log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct",
             "log_1_min_cumprod_ct")
    return {n: (tab[i, :T] if i < 4 else tab[i]) for i, n in enumerate(names)}


def gemm(A, W, C_out, M, N, K, *, bias=None, R=None, lda=None, ldw=None, ldc=None, ldr=None, groups=1, a_gstride=0,
         w_gstride=0, c_gstride=0, act=0, split2=None, w_plane=None, **unused):
    """_lib.gemm for the dense loader: C = act(out_scale * A W^T + bias) + R per group (ds_gemm / ds_gemm_f16x2)."""
    for k, v in unused.items():
        assert not v or k in ("loader", "pro", "store", "dil"), "emulation: unsupported gemm option %s=%r" % (k, v)
    assert K % 32 == 0, "K must be a positive multiple of 32"
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    ldc = N if ldc is None else ldc
    ldr = ldc if ldr is None else ldr
    Af, Cf = _flat(A), _flat(C_out)
    if split2 is not None:
        assert groups <= 1 or (bias is None and R is None), "groups: no bias / residual"
        assert W.dtype == torch.int16
        Wf = _flat(W).view(torch.float16)
        plane = N * ldw if w_plane is None else w_plane
    else:
        Wf = _flat(W)
    for g in range(max(1, groups)):
        a = Af[g * a_gstride:].as_strided((M, K), (lda, 1))
        if split2 is not None:
            hi, lo = _split16(a)
            a = hi.float() + lo.float()
            w = (Wf[g * w_gstride:].as_strided((N, K), (ldw, 1)).float() +
                 Wf[plane + g * w_gstride:].as_strided((N, K), (ldw, 1)).float())
            res = (a.double() @ w.double().t()).float() * split2
        else:
            w = Wf[g * w_gstride:].as_strided((N, K), (ldw, 1))
            res = (a.double() @ w.double().t()).float()
        if bias is not None:
            res = res + bias
        if act == 1:
            res = res * torch.sigmoid(1.702 * res)
        if R is not None:
            res = res + _flat(R)[g * c_gstride:].as_strided((M, N), (ldr, 1))
        Cf[g * c_gstride:].as_strided((M, N), (ldc, 1)).copy_(res)
    return C_out


class Lib:
    """the entries of libdiffsound_hip.so that modeling/train.py, AdaLayerNorm.table and q_sample_tokens reach"""

    def ds_colsum(self, x, out, G, R, C, ld, gstride, accumulate, stream):
        xf, of = _flat(x), _flat(out)
        for g in range(G):
            s = xf[g * gstride:].as_strided((R, C), (ld, 1)).sum(0)
            of[g * C:(g + 1) * C] = of[g * C:(g + 1) * C] + s if accumulate else s
        return 0

    def ds_colsum_ws(self, x, out, G, R, C, ld, gstride, accumulate, work, work_floats, stream):
        assert work is not None and work_floats >= G * 64 * C          # the documented always-enough size
        return self.ds_colsum(x, out, G, R, C, ld, gstride, accumulate, stream)

    def ds_convert_operand(self, src, rows, cols, ld_src, transpose, scale, dst, ld_dst, plane, dst_f16, stream):
        assert ld_dst % 8 == 0 and ld_dst >= (rows if transpose else cols) and ld_src >= cols
        v = _flat(src).as_strided((rows, cols), (ld_src, 1)) * scale
        v = v.t() if transpose else v
        drows, dvalid = v.shape
        full = torch.zeros(drows, ld_dst)
        full[:, :dvalid] = v
        if dst_f16:
            assert dst.dtype == torch.int16 and plane >= drows * ld_dst
            hi, lo = _split16(full)
            df = _flat(dst).view(torch.float16)
            df[:drows * ld_dst] = hi.reshape(-1)
            df[plane:plane + drows * ld_dst] = lo.reshape(-1)
        else:
            _flat(dst)[:drows * ld_dst] = full.reshape(-1)
        return 0

    def ds_amax(self, x, n, out, stream):
        out[0] = torch.maximum(out[0], _flat(x)[:n].abs().max())
        return 0

    def ds_adaln(self, x, y, M, L, D, table, t, stream):
        xn = torch.nn.functional.layer_norm(x.view(M, D), (D,), eps=1e-5)
        e = table[t]                                                    # [B, 2D]
        B = M // L
        y.view(B, L, D).copy_(xn.view(B, L, D) * (1 + e[:, None, :D]) + e[:, None, D:])
        return 0

    def ds_layernorm(self, x, y, M, D, gamma, beta, stream):
        y.view(M, D).copy_(torch.nn.functional.layer_norm(x.view(M, D), (D,), gamma, beta, eps=1e-5))
        return 0

    def ds_layernorm_bwd(self, x, dy, dx, dyxn, M, L, D, mode, table, t, gamma, stream):
        x, dy = x.view(M, D), dy.view(M, D)
        mean = x.mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((x - mean) ** 2).mean(1, keepdim=True) + 1e-5)
        xn = (x - mean) * rstd
        if mode == 0:
            s = (1 + table[t][:, :D]).repeat_interleave(L, dim=0)
        else:
            s = gamma[None, :]
        g = dy * s
        dx.view(M, D).copy_(rstd * (g - g.mean(1, keepdim=True) - xn * (g * xn).mean(1, keepdim=True)))
        if dyxn is not None:
            dyxn.view(M, D).copy_(dy * xn)
        return 0

    def ds_gelu2(self, x, dy, out, n, stream):
        xf = _flat(x)[:n]
        sg = torch.sigmoid(1.702 * xf)
        _flat(out)[:n] = xf * sg if dy is None else _flat(dy)[:n] * (sg + 1.702 * xf * sg * (1 - sg))
        return 0

    def ds_softmax_rows(self, x, rows, n, ld, scale, stream):
        v = _flat(x).as_strided((rows, ld), (ld, 1))
        p = torch.softmax(v[:, :n] * scale, dim=1)
        v.zero_()
        v[:, :n] = p
        return 0

    def ds_softmax_bwd_rows(self, P, dP, rows, n, ld, scale, stream):
        p = _flat(P).as_strided((rows, ld), (ld, 1))[:, :n]
        d = _flat(dP).as_strided((rows, ld), (ld, 1))
        ds = scale * p * (d[:, :n] - (d[:, :n] * p).sum(1, keepdim=True))
        d.zero_()
        d[:, :n] = ds
        return 0

    @staticmethod
    def _heads(x, ld, B, L, H):
        """flat storage starting at the operand's first column, row stride ld -> [B, H, L, 64] view"""
        return _flat(x).as_strided((B, H, L, 64), (L * ld, 64, ld, 1))

    def ds_attention(self, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Lq, Lk, scale, stream):
        Q, K, V = self._heads(q, ldq, B, Lq, H), self._heads(k, ldk, B, Lk, H), self._heads(v, ldv, B, Lk, H)
        P = torch.softmax(scale * (Q.double() @ K.double().transpose(-1, -2)), dim=-1)
        self._heads(o, ldo, B, Lq, H).copy_((P @ V.double()).float())
        return 0

    def ds_attention_bwd(self, q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, dq, lddq, dk, lddk, dv, lddv, stats, B, H, Lq, Lk,
                         scale, stream):
        assert stats.numel() >= 2 * B * H * ((Lq + 31) // 32 * 32)
        with torch.enable_grad():
            Q = self._heads(q, ldq, B, Lq, H).double().requires_grad_(True)
            K = self._heads(k, ldk, B, Lk, H).double().requires_grad_(True)
            V = self._heads(v, ldv, B, Lk, H).double().requires_grad_(True)
            out = torch.softmax(scale * (Q @ K.transpose(-1, -2)), dim=-1) @ V
            out.backward(self._heads(d_o, lddo, B, Lq, H).double())
        self._heads(dq, lddq, B, Lq, H).copy_(Q.grad.float())
        self._heads(dk, lddk, B, Lk, H).copy_(K.grad.float())
        self._heads(dv, lddv, B, Lk, H).copy_(V.grad.float())
        return 0

    def ds_embed(self, tok, emb, pos, out, M, L, D, stream):
        out.view(M, D).copy_(emb[tok.reshape(-1)] + pos.repeat(M // L, 1))
        return 0

    def ds_embed_bwd(self, dx, tok, demb, M, D, rows, stream):
        demb.index_add_(0, tok.reshape(-1), dx.view(M, D))
        return 0

    def ds_axpy(self, y, x, a, n, stream):
        _flat(y)[:n] += a * _flat(x)[:n]
        return 0

    def ds_q_sample(self, x0, t, u, sched, out, B, L, K, T, stream):
        out.copy_(O.q_sample(_sched_dict(sched, T), x0, t, u, K + 1).argmax(1))
        return 0

    def ds_loss_tail(self, logits, x0, xt, t, sched, kl, nll, kl_aux, dbg, B, L, K, T, stream):
        sd = _sched_dict(sched, T)
        lg = logits.view(B, L, K).permute(0, 2, 1)                      # the oracle's [B, K, L]
        log_x_start, log_xt = O.log_onehot(x0, K + 1), O.log_onehot(xt, K + 1)
        recon = O.predict_start(lg)
        model = O.q_posterior(sd, recon, log_xt, t)
        true = O.q_posterior(sd, log_x_start, log_xt, t)
        kl_of = lambda a, b: (a.exp() * (a - b)).sum(dim=1)
        kl.copy_(kl_of(true, model))
        nll.copy_(-(log_x_start.exp() * model).sum(dim=1))
        kl_aux.copy_(kl_of(log_x_start[:, :-1], recon[:, :-1]))
        return 0

    def ds_loss_tail_bwd(self, logits, x0, xt, t, pt, sched, dlogits, B, L, K, T, mw0, mw1, aux_w, adaptive, stream):
        lg = logits.view(B, L, K).permute(0, 2, 1)
        d = O.loss_tail_backward(_sched_dict(sched, T), lg, x0, xt, t, pt, num_timesteps=T, mask_weight=(mw0, mw1),
                                 auxiliary_loss_weight=aux_w, adaptive_auxiliary_loss=bool(adaptive))
        dlogits.view(B, L, K).copy_(d.permute(0, 2, 1))
        return 0

    def ds_adamw(self, p, g, m, v, n, lr, b1, b2, eps, wd, step, stream):
        return self._adamw(p, g, m, v, lr, 1 - b1 ** step, math.sqrt(1 - b2 ** step), 1.0, b1, b2, eps, wd)

    def ds_adamw_dev(self, p, g, m, v, n, hyper, b1, b2, eps, wd, stream):
        lr, bc1, bc2s, gs = (float(h) for h in hyper)
        return self._adamw(p, g, m, v, lr, bc1, bc2s, gs, b1, b2, eps, wd)

    @staticmethod
    def _adamw(p, g, m, v, lr, bc1, bc2s, gs, b1, b2, eps, wd):
        g = g * gs
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.mul_(1 - lr * wd).sub_((lr / bc1) * m / (v.sqrt() / bc2s + eps))
        return 0

    def ds_denoiser_destroy(self, h):
        return None


def install(monkeypatch):
    """Route text_to_sound_synthesis_amd._lib to the emulation for the duration of a test."""
    from text_to_sound_synthesis_amd import _lib
    fake = Lib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(_lib, "ptr", lambda t: t)
    monkeypatch.setattr(_lib, "ptr_off", lambda t, elems: _flat(t)[elems:])
    monkeypatch.setattr(_lib, "stream", lambda: None)
    monkeypatch.setattr(_lib, "gemm", gemm)
    return fake
