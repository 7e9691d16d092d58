"""Training step of the denoiser on the HIP kernels (scope row 8f-3).

    loss, grads = TrainStep(model.transformer).loss_and_grads(x0, cond_emb, t, pt, noise)

follows DiffusionTransformer._train_loss / forward (diffusion_transformer.py:408-476,539-577) and what
`loss.backward()` produces for every parameter of `Text2ImageTransformer` (engine/solver_spec.py:308-331 back-propagates
exactly this), without an autograd graph: the forward keeps the activations the backward needs, every nn.Linear
contributes three GEMMs (y = x W^T, dX = dY W, dW = dY^T X -- 98 % of the step's flops), the row / elementwise
pieces run on csrc/train.hip, norm.hip and sampler.hip.  The self-attention Q | K | V projections and the
cross-attention K | V projections are fused into one linear each (one GEMM of N = 3D / 2D instead of three / two).

Two GEMM backends (`TrainStep(precision=...)`), both behind the same step:

  "fp32"   exact-fp32 MFMA (`ds_gemm`); transposed / padded operands are made by torch.  The reference arithmetic.
  "f16x2"  the fp32-class 3-pass fp16 split GEMM with PACKED split planes on both operands of all three GEMMs (LDS-DMA
           staging, gemm_f16x2.hip AMODE 2: 1.2-1.4x the loader-split program on the step's shapes) and NOTHING on the host
           between two launches.  Every matrix that enters a GEMM goes through `ds_pack_operand` exactly once (csrc/pack.hip):
             * an activation x -> its row form (A of y = x W^T) and its transposed form (X^T, W operand of dW = dY^T X);
             * a gradient dY  -> its row form (A of dX = dY W), its transposed form (A of dW), the per-tile column sums that
               become the bias gradient, and max |dY| for the loss-scale monitor -- one read of dY instead of four;
             * a weight W     -> row form (forward) and transposed form (dX), with a per-matrix power-of-two pre-scale
               refreshed every `rescale_interval` steps (weights move slowly);
             * GELU2 rides in the pack's prologue in both directions (fc2's input from fc1's output; d fc1-output from
               d gelu-output), so the MLP's activation and its gradient never exist in fp32.
           dW runs as a split-K launch on the packed planes (`groups` K-ranges fill the chip: a 1024 x 1024 dW is only 64
           tiles) whose partial sums `ds_colsum` adds in a fixed order.  The gradients' magnitude (|dY| ~ 1e-12 .. 1e-2, far
           below fp16's range, and spread over 2^25 between the sites of one backward) is handled by a loss scale 2^k -- d
           logits is multiplied by it, every dX carries it, the dW GEMMs' epilogue and one multiply over the small gradients
           take it out again -- PLUS one power of two per site (every linear's dY, every attention backward's dO; round 6): the
           operand times 2^e is what is split to fp16 and 2^-e goes into the consuming epilogues / stores, exact.  k and the e
           come from a two-pass calibration (`calibrate`: max |operand| per site, two host syncs).  A split value keeps an
           absolute precision of 2^-25, so anything above 2^-3 after scaling is fp32-class; the calibration puts every site's
           largest value at 2^6..2^7 under a saturation monitor whose window ends at 2^15.  The step has no host
           synchronisation at all and is captured in a hipGraph (`capture`): one graph launch per iteration instead of ~2000
           kernel launches from Python.

Attention: `attention="fused"` (default) = a fused forward + a backward by tile-wise recomputation that reads Q | K | V and
writes dQ | dK | dV in place in the fused projection buffers and never stores the probabilities -- `ds_attention` +
`ds_attention_bwd` on the exact-fp32 MFMA in the "fp32" backend, `ds_attention_f16x2` + `ds_attention_bwd_f16x2` (every tile
product on the fp16 matrix cores, 3-pass split, fp32-class) in the "f16x2" backend; `attention="composed"` = grouped fp32
GEMMs + row softmax with materialised probabilities and torch head split / merge / transposes (the first version, kept as a
cross-check).  The norms and the loss tail are exact fp32.  Worst per-tensor gradient error against autograd through the oracle, all 63 tensors of the
2-layer test model: see tests/test_hip_train_kernels.py.
"""
import math

import torch

from .. import _lib

L_ = _lib


def _ceil(a, b):
    return (a + b - 1) // b * b


def _colsum(x, G=1, R=None, accumulate_into=None):
    """out[g][c] = sum over the R rows of group g (tall inputs: chunked two-stage sum, ds_colsum_ws)"""
    M, C_ = x.shape
    R = M // G if R is None else R
    out = torch.empty(G, C_, device=x.device) if accumulate_into is None else accumulate_into
    if R >= 64:
        work = torch.empty(G * 64 * C_, device=x.device)
        L_.check(L_.lib().ds_colsum_ws(L_.ptr(x), L_.ptr(out), G, R, C_, C_, R * C_, int(accumulate_into is not None),
                                       L_.ptr(work), work.numel(), L_.stream()))
    else:
        L_.check(L_.lib().ds_colsum(L_.ptr(x), L_.ptr(out), G, R, C_, C_, R * C_, int(accumulate_into is not None), L_.stream()))
    return out


PACK_PLAIN, PACK_GELU2, PACK_GELU2_BWD = 0, 1, 2


class _Linear:
    """One (possibly fused) nn.Linear of the step: the weights of its parts (each [N_i][K] fp32; query | key | value of a fused
    projection), bias [N], plus what the GEMM backend derived from them.  `W` (the concatenated fp32 matrix) is only built
    when somebody asks for it -- the "fp32" backend; the "f16x2" backend packs every part straight into its range of the
    fused operand (ds_pack_operand's sub-range form)."""

    def __init__(self, key, W, b):
        self.key = key
        self.parts = [w.detach() for w in W] if isinstance(W, (list, tuple)) else [W.detach()]
        self.b = torch.cat([x.detach() for x in b]) if isinstance(b, (list, tuple)) else b.detach()
        self.N, self.K = sum(w.shape[0] for w in self.parts), self.parts[0].shape[1]
        self._W = self.parts[0] if len(self.parts) == 1 else None
        self.extra = {}

    @property
    def W(self):
        if self._W is None:
            self._W = torch.cat(self.parts)
        return self._W


def _gelu2(x, dy=None):
    out = torch.empty_like(x)
    L_.check(L_.lib().ds_gelu2(L_.ptr(x), L_.ptr(dy), L_.ptr(out), x.numel(), L_.stream()))
    return out


class _Fp32Gemm:
    """Backend "fp32": every GEMM on the exact-fp32 MFMA kernel; transposes and zero padding by torch.  An operand handle is
    the fp32 matrix itself (after the elementwise prologue, if any)."""
    name = "fp32"

    def prepare(self, lin):
        pass

    def prep_x(self, lin, x, pro=PACK_PLAIN):
        return _gelu2(x) if pro == PACK_GELU2 else x

    def prep_dy(self, lin, dy, pro=PACK_PLAIN, aux=None, amax=None, need_row=True, scale=1.0):
        return _gelu2(aux, dy) if pro == PACK_GELU2_BWD else dy

    def fwd(self, lin, x, R=None):
        M = x.shape[0]
        y = torch.empty(M, lin.N, device=x.device)
        return L_.gemm(x, lin.W, y, M, lin.N, lin.K, bias=lin.b, R=R)

    def dx(self, lin, dy, unscale=1.0):
        M, N, K = dy.shape[0], lin.N, lin.K
        Np = _ceil(N, 32)
        Wt = lin.extra.get("Wt")
        if Wt is None:                                  # [K][Np]: rows are K-contiguous operands of the GEMM
            Wt = torch.zeros(K, Np, device=dy.device)
            Wt[:, :N] = lin.W.t()
            lin.extra["Wt"] = Wt
        dyp = dy if Np == N else torch.nn.functional.pad(dy, (0, Np - N))
        out = torch.empty(M, K, device=dy.device)
        return L_.gemm(dyp.contiguous(), Wt, out, M, K, Np)

    def dw(self, lin, x, dy, inv_scale):
        M = x.shape[0]
        Mp = _ceil(M, 32)

        def pad_t(a):                                   # a [M][C] -> a^T zero-padded to [C][Mp]
            out = torch.zeros(a.shape[1], Mp, device=a.device)
            out[:, :M] = a.t()
            return out
        dW = torch.empty(lin.N, lin.K, device=x.device)
        L_.gemm(pad_t(dy), pad_t(x), dW, lin.N, lin.K, Mp)
        if inv_scale != 1.0:
            dW.mul_(inv_scale)
        return dW

    def db(self, lin, dy):
        return _colsum(dy)[0]


class _Packed:
    """What ds_pack_operand made of one fp32 matrix [rows][cols]: `row` = packed planes of the matrix (int16 [2][plane]),
    `t` = packed planes of its transpose with the contraction index padded to rows_pad, `part` = per-64-row column sums."""
    __slots__ = ("rows", "cols", "row", "row_plane", "t", "t_plane", "rows_pad", "part")


def _pack(src, rows, cols, *, scale=1.0, pro=PACK_PLAIN, aux=None, want_row=True, rows_pad=0, colsum=False, amax=None, ld=None):
    """One ds_pack_operand launch (csrc/pack.hip) over src [rows][ld >= cols]."""
    dev = src.device
    o = _Packed()
    o.rows, o.cols, o.rows_pad = rows, cols, rows_pad
    o.row = o.t = o.part = None
    o.row_plane = _ceil(rows, 16) * cols
    o.t_plane = _ceil(cols, 16) * rows_pad
    if want_row:
        o.row = torch.empty(2, o.row_plane, dtype=torch.int16, device=dev)
    if rows_pad:
        o.t = torch.empty(2, o.t_plane, dtype=torch.int16, device=dev)
    if colsum:
        o.part = torch.empty(L_.lib().ds_pack_operand_tile_rows(rows, rows_pad), cols, device=dev)
    L_.check(L_.lib().ds_pack_operand(L_.ptr(src), rows, cols, cols if ld is None else ld, float(scale), int(pro), L_.ptr(aux),
                                      cols, L_.ptr(o.row), o.row_plane, L_.ptr(o.t), o.t_plane, rows_pad, 0, 0, L_.ptr(o.part),
                                      L_.ptr(amax), L_.stream()))
    return o


def _pack_parts(parts, K, scale):
    """The parts [N_i][K] of a fused weight (N = sum N_i, every N_i % 32 == 0) -> ONE _Packed of the fused matrix [N][K]: part i
    goes to the row groups [n0 / 16, ..) of the row form and to the k-range [n0, n0 + N_i) of the transposed form [K][N]."""
    dev = parts[0].device
    N = sum(w.shape[0] for w in parts)
    assert all(w.shape[0] % 32 == 0 and w.shape[1] == K and w.is_contiguous() for w in parts)
    o = _Packed()
    o.rows, o.cols, o.rows_pad, o.part = N, K, N, None
    o.row_plane, o.t_plane = N * K, _ceil(K, 16) * N
    o.row = torch.empty(2, o.row_plane, dtype=torch.int16, device=dev)
    o.t = torch.empty(2, o.t_plane, dtype=torch.int16, device=dev)
    n0 = 0
    for w in parts:
        Ni = w.shape[0]
        L_.check(L_.lib().ds_pack_operand(L_.ptr(w), Ni, K, K, float(scale), PACK_PLAIN, None, 0,
                                          L_.ptr_off(o.row, n0 * K), o.row_plane,       # row group n0 / 16: (n0 / 16) * (K / 32) * 512 halves
                                          L_.ptr(o.t), o.t_plane, Ni, n0, N, None, None, L_.stream()))
        n0 += Ni
    return o


class _SplitGemm:
    """Backend "f16x2": every linear-layer GEMM on the 3-pass fp16 split kernel with packed split planes on both operands
    (module docstring).  Operand handles are `_Packed` objects."""
    name = "f16x2"

    def __init__(self):
        self.wexp = {}              # key -> s: the weight is split as W * 2^s (max |W| 2^s in [2^13, 2^14))
        self.rows_per_sample = 0    # token rows per sample of the activations (set by the step; 0: unknown)

    @staticmethod
    def scales_of(lins):
        """{key: s} with s = 13 - floor(log2 max|W|) for every matrix; one host sync for all of them"""
        mx = torch.stack([torch.stack([w.abs().max() for w in l.parts]).max() for l in lins]).tolist()
        return {l.key: 0 if (m == 0.0 or not math.isfinite(m)) else 13 - math.floor(math.log2(m)) for l, m in zip(lins, mx)}

    def refresh_scales(self, lins):
        self.wexp.update(self.scales_of(lins))

    @staticmethod
    def split_k(N, K, M=4096):
        """K-ranges of a dW = dY^T X launch over M rows, from the measured sweep of the packed kernel INCLUDING the fixed-order
        reduction of the partial results (tools/train_gemm_ab.py -> profiles/r05g_train_gemm_packed_sweep.txt, M = 5300 / 1540):
        >= 256 tiles of 128 x 128 (fc1 / fc2: half a round of the chip) run unsplit -- the partials' write + re-read costs more
        than the idle slots (133 vs 148 us); so does 3072 x 1024 since round 6 (256 tiles of 96 x 128, one per CU: 111 us against
        118 in 4 ranges, profiles/r06x_train_gemm_tile_sweep.txt); smaller products take 4 ranges (1024 x 1024: 48 us against 52
        at 8, 83 unsplit), 2 when the contraction itself is short (the caption rows: 22 us against 31 at 8)."""
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        if tiles >= 192:
            return 1
        return 4 if M >= 4096 else 2

    def rows_pad(self, lin, M):
        """the padded contraction length of this layer's dW = dY^T X over M rows: a multiple of 32 per K-range"""
        return _ceil(M, 32 * self.split_k(lin.N, lin.K, M))

    def prepare(self, lin):
        """W * 2^s -> row form [N][K] (forward) and transposed form [K][ceil32(N)] (dX), one pass"""
        s = self.wexp[lin.key]
        lin.extra["osc"] = 2.0 ** (-s)
        if len(lin.parts) > 1 or lin.N % 32 == 0:
            lin.extra["Wp"] = _pack_parts(lin.parts, lin.K, 2.0 ** s)
        else:
            lin.extra["Wp"] = _pack(lin.W, lin.N, lin.K, scale=2.0 ** s, rows_pad=_ceil(lin.N, 32))

    def prep_x(self, lin, x, pro=PACK_PLAIN):
        M = x.shape[0]
        return _pack(x, M, lin.K, pro=pro, rows_pad=self.rows_pad(lin, M))

    def prep_dy(self, lin, dy, pro=PACK_PLAIN, aux=None, amax=None, need_row=True, scale=1.0):
        """scale: the site's own power of two (TrainStep._site_exp): the planes hold dY * scale, the column sums (bias
        gradient) are those of dY itself, `amax` takes max |dY * scale|"""
        M = dy.shape[0]
        return _pack(dy, M, lin.N, scale=scale, pro=pro, aux=aux, want_row=need_row, rows_pad=self.rows_pad(lin, M), colsum=True,
                     amax=amax)

    def fwd(self, lin, xp, R=None):
        M = xp.rows
        y = torch.empty(M, lin.N, device=xp.row.device)
        Wp = lin.extra["Wp"]
        # (rows_per_sample: lets the dispatcher take the sampling loop's per-sample 272 x 256 program where its grid pays --
        #  the 20 x 12 tiles of the QKV projection, 91 us against 101; same bits, tests/test_hip_widening.py)
        return L_.gemm(xp.row, Wp.row, y, M, lin.N, lin.K, bias=lin.b, R=R, split2=lin.extra["osc"], a_plane=xp.row_plane,
                       w_plane=Wp.row_plane, rows_per_sample=self.rows_per_sample if M % max(1, self.rows_per_sample) == 0 else 0)

    def dx(self, lin, dyp, unscale=1.0):
        """unscale: 2^-e of the site's own scale, folded into the epilogue's output scale (exact: powers of two)"""
        M = dyp.rows
        Wp = lin.extra["Wp"]
        out = torch.empty(M, lin.K, device=dyp.row.device)
        Np = Wp.rows_pad                                       # contraction length of dX = dY W (N, a multiple of 32 here)
        assert Np == lin.N, "dX needs N % 32 == 0 (true for every linear of this network)"
        return L_.gemm(dyp.row, Wp.t, out, M, lin.K, Np, split2=lin.extra["osc"] * unscale, a_plane=dyp.row_plane,
                       w_plane=Wp.t_plane, rows_per_sample=self.rows_per_sample if M % max(1, self.rows_per_sample) == 0 else 0)

    def dw(self, lin, xp, dyp, inv_scale, out=None):
        N, K, Mp = lin.N, lin.K, dyp.rows_pad
        assert xp.rows_pad == Mp and xp.cols == K and dyp.cols == N
        S = self.split_k(N, K, dyp.rows)
        dev = dyp.t.device
        dW = torch.empty(N, K, device=dev) if out is None else out
        if S == 1:
            return L_.gemm(dyp.t, xp.t, dW, N, K, Mp, split2=inv_scale, a_plane=dyp.t_plane, w_plane=xp.t_plane)
        part = torch.empty(S, N * K, device=dev)
        Kc = Mp // S
        L_.gemm(dyp.t, xp.t, part, N, K, Kc, lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16,
                c_gstride=N * K, split2=inv_scale, a_plane=dyp.t_plane, w_plane=xp.t_plane)
        L_.check(L_.lib().ds_colsum(L_.ptr(part), L_.ptr(dW), 1, S, N * K, N * K, 0, 0, L_.stream()))
        return dW

    def dw_many(self, items):
        """items: [(lin, xp, dyp, inv_scale, dW)] -- the weight gradients of several layers whose operands are all packed, into
        the pre-allocated dW tensors.  Every dW launch is sized to about one workgroup per CU, and a workgroup alone on a CU
        runs its tile in ~0.6 of the time two co-resident ones take: products of equal tile configuration and K-range count go
        out as ONE grid (ds_gemm_f16x2_multi, up to four), the same bits as one launch each."""
        groups = {}
        for it in items:
            lin, xp, dyp = it[0], it[1], it[2]
            S = self.split_k(lin.N, lin.K, dyp.rows)
            cfg = L_.lib().ds_gemm_f16x2_auto_tile(lin.N, lin.K, S)
            groups.setdefault((cfg, S), []).append(it)
        for (cfg, S), its in groups.items():
            for c0 in range(0, len(its), 4):
                chunk = its[c0:c0 + 4]
                if cfg == 2 or len(chunk) == 1:
                    for lin, xp, dyp, inv_scale, dW in chunk:
                        self.dw(lin, xp, dyp, inv_scale, out=dW)
                    continue
                descs, finish = [], []
                for lin, xp, dyp, inv_scale, dW in chunk:
                    N, K, Mp = lin.N, lin.K, dyp.rows_pad
                    assert xp.rows_pad == Mp and xp.cols == K and dyp.cols == N
                    if S == 1:
                        descs.append(L_.gemm(dyp.t, xp.t, dW, N, K, Mp, split2=inv_scale, a_plane=dyp.t_plane, w_plane=xp.t_plane,
                                             desc_only=True))
                    else:
                        part = torch.empty(S, N * K, device=dW.device)
                        Kc = Mp // S
                        descs.append(L_.gemm(dyp.t, xp.t, part, N, K, Kc, lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc * 16,
                                             w_gstride=Kc * 16, c_gstride=N * K, split2=inv_scale, a_plane=dyp.t_plane,
                                             w_plane=xp.t_plane, desc_only=True))
                        finish.append((part, dW, N * K))
                L_.gemm_multi(descs, cfg)
                for part, dW, n in finish:
                    L_.check(L_.lib().ds_colsum(L_.ptr(part), L_.ptr(dW), 1, S, n, n, 0, 0, L_.stream()))

    def db(self, lin, dyp):
        out = torch.empty(1, dyp.cols, device=dyp.part.device)
        R = (dyp.rows + 63) // 64                              # tile rows that hold data (the rest pad the contraction)
        L_.check(L_.lib().ds_colsum(L_.ptr(dyp.part), L_.ptr(out), 1, R, dyp.cols, dyp.cols, 0, 0, L_.stream()))
        return out[0]


def _norm_fwd(x, mode, L, table=None, t=None, gamma=None, beta=None):
    M, D = x.shape
    y = torch.empty_like(x)
    if mode == 0:
        L_.check(L_.lib().ds_adaln(L_.ptr(x), L_.ptr(y), M, L, D, L_.ptr(table), L_.ptr(t), L_.stream()))
    else:
        L_.check(L_.lib().ds_layernorm(L_.ptr(x), L_.ptr(y), M, D, L_.ptr(gamma), L_.ptr(beta), L_.stream()))
    return y


def _norm_bwd(x, dy, mode, L, table=None, t=None, gamma=None, add_to=None):
    """-> (dx, d scale, d shift): per sample [B][D] (AdaLN, mode 0) or summed [1][D] (LayerNorm, mode 1).  One kernel computes dx
    and per-chunk column sums of dy * xn and dy (ds_layernorm_bwd_sums), one ds_colsum adds the chunks.
    add_to: the residual stream's gradient -- the norm-input gradient is ADDED to it in place instead of being returned."""
    M, D = x.shape
    dx = torch.empty_like(x) if add_to is None else add_to
    G = M // L if mode == 0 else 1
    chunks = L_.lib().ds_layernorm_bwd_chunks(M, L, mode)
    part = torch.empty(G * chunks, 2 * D, device=x.device)
    L_.check(L_.lib().ds_layernorm_bwd_sums(L_.ptr(x), L_.ptr(dy), L_.ptr(dx), L_.ptr(part), M, L, D, mode, L_.ptr(table), L_.ptr(t),
                                            L_.ptr(gamma), int(add_to is not None), L_.stream()))
    sums = torch.empty(G, 2 * D, device=x.device)
    L_.check(L_.lib().ds_colsum(L_.ptr(part), L_.ptr(sums), G, chunks, 2 * D, 2 * D, chunks * 2 * D, 0, L_.stream()))
    return dx, sums[:, :D], sums[:, D:]


def _heads(x, B, Lx, H, Lp):
    """[B*Lx, H*64] (a column range of a fused projection is fine: any row stride) -> [B*H, Lp, 64] zero-padded"""
    out = torch.zeros(B * H, Lp, 64, device=x.device)
    out.view(B, H, Lp, 64)[:, :, :Lx].copy_(x.view(B, Lx, H, 64).permute(0, 2, 1, 3))
    return out


def _merge_into(out, x4, B, Lx, H):
    """[B*H, Lp, 64] -> out [B*Lx, H*64] (any row stride)"""
    out.view(B, Lx, H, 64).copy_(x4.view(B, H, -1, 64)[:, :, :Lx].permute(0, 2, 1, 3))   # .view: never a silent copy
    return out


class _Attn:
    """softmax(q k^T / 8) v per head (FullAttention / CrossAttention cores, transformer_utils.py:43-58,91-109)"""

    def __init__(self, q, k, v, B, Lq, Lk, H):
        self.B, self.Lq, self.Lk, self.H = B, Lq, Lk, H
        self.Lqp, self.Lkp = _ceil(Lq, 32), _ceil(Lk, 32)
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
        self.out = _merge_into(torch.empty(B * Lq, H * 64, device=q.device), o4, B, Lq, H)

    def backward(self, dO, dq_out, dk_out, dv_out):
        """dO [B*Lq, H*64] -> writes dQ / dK / dV into the given [rows, H*64] views (column ranges of a fused gradient)"""
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
        _merge_into(dq_out, dQ4, B, Lq, H)
        _merge_into(dk_out, dK4, B, Lk, H)
        _merge_into(dv_out, dV4, B, Lk, H)


class _FusedAttn:
    """The same attention cores on the fused kernels: `ds_attention` forward and `ds_attention_bwd` (tile-wise
    recomputation: the probabilities are never stored), reading Q / K / V and writing dQ / dK / dV IN PLACE in the fused
    projection buffers -- an operand is (tensor, first column, row stride).  Exact-fp32 MFMA, like the composed version."""

    def __init__(self, q, k, v, B, Lq, Lk, H, split=False):
        """split: the forward on the streamed fp16-split attention kernel of the sampling path (ds_attention_f16x2: fp32-class,
        2e-5 max-abs against float64 where the exact-fp32 kernel has 1e-6; 3-4x faster) -- the "f16x2" backend's choice; the
        backward recomputes the probabilities in exact fp32 either way."""
        self.q, self.k, self.v, self.B, self.Lq, self.Lk, self.H, self.split = q, k, v, B, Lq, Lk, H, split
        self.out = torch.empty(B * Lq, H * 64, device=q[0].device)
        fn = L_.lib().ds_attention_f16x2 if split else L_.lib().ds_attention
        L_.check(fn(L_.ptr_off(q[0], q[1]), q[2], L_.ptr_off(k[0], k[1]), k[2], L_.ptr_off(v[0], v[1]), v[2],
                    L_.ptr(self.out), H * 64, B, H, Lq, Lk, 0.125, L_.stream()))

    def backward(self, dO, dq, dk, dv, amax=None, do_scale=1.0):
        """split: the tile products of the recomputation on the fp16 matrix cores too (ds_attention_bwd_f16x2_mon); dO enters
        them as an fp16-split operand under the step's loss scale AND this call's own power of two `do_scale` (the kernel
        splits dO * do_scale and takes it out again at its stores), and the kernel folds max |dO * do_scale| into the saturation
        monitor (`amax`) itself; the in-register dS is normalised per wave inside the kernel (csrc/attention_bwd.hip)."""
        q, k, v, B, Lq, Lk, H = self.q, self.k, self.v, self.B, self.Lq, self.Lk, self.H
        stats = torch.empty(2 * B * H * _ceil(Lq, 32), device=dO.device)
        args = (L_.ptr_off(q[0], q[1]), q[2], L_.ptr_off(k[0], k[1]), k[2], L_.ptr_off(v[0], v[1]), v[2], L_.ptr(self.out), H * 64,
                L_.ptr(dO), H * 64, L_.ptr_off(dq[0], dq[1]), dq[2], L_.ptr_off(dk[0], dk[1]), dk[2], L_.ptr_off(dv[0], dv[1]), dv[2],
                L_.ptr(stats), B, H, Lq, Lk, 0.125)
        if self.split and amax is not None:
            L_.check(L_.lib().ds_attention_bwd_f16x2_mon(*args, float(do_scale), L_.ptr(amax), L_.stream()))
        else:
            L_.check((L_.lib().ds_attention_bwd_f16x2 if self.split else L_.lib().ds_attention_bwd)(*args, L_.stream()))


@torch.no_grad()
def training_inputs(model, batch, generator=None):
    """The reference's training batch -> the five tensors of `TrainStep.loss_and_grads`: what happens between
    `self.model(batch, return_loss=True)` (engine/solver_spec.py:308-331) and the loss arithmetic.

        batch {'image': mel f32[B,1,80,848], 'text': list of B captions}
          -> DALLE.prepare_input (dalle_spec.py:93-133): Tokenize (BPE ids i64[B,77], host) and VQModel.encode + argmin +
             ColumnMajor of the mel -> content_token i64[B,265]                       (HIP: vqgan.py, ds_vq_argmin)
          -> DiffusionTransformer.forward (diffusion_transformer.py:552-566): CLIPTextEmbedding of the ids -> f32[B,77,512]
          -> _train_loss (:408-421): sample_time(b, device, 'importance') -> (t, pt); the uniforms q_sample's
             log_sample_categorical draws as torch.rand_like(logits [B, K+1, L]) (:359-368)

    `model` is the DALLE drop-in (`condition_codec` + `transformer.condition_emb` built: config.default_config(with_clip=True)).
    A batch that already carries 'condition_embed_token' / 'condition_token' (DALLE.prepare_condition's other forms) or
    'content_token' skips the respective stage.  generator: torch.Generator of the model's device for t and the noise.
    Everything is enqueued on the current stream; the only host synchronisation is sample_time's Lt_count test while it
    is still false."""
    x0, cond_emb = training_prologue(model, batch)
    return training_draws(model, x0, cond_emb, generator=generator)


@torch.no_grad()
def training_prologue(model, batch):
    """The part of `training_inputs` that depends on the batch and on FROZEN weights only (BPE, the CLIP text tower, the VQ
    encoder): (x0 i64[B,265], cond_emb f32[B,77,512]).  Nothing the optimiser touches is read, so it may run ahead of the
    previous iteration -- GraphSolver.prefetch enqueues it on a side stream."""
    dt = model.transformer
    dev = dt.device
    cond = model.prepare_condition(batch)
    if batch.get("content_token") is not None:
        x0 = batch["content_token"].to(dev)
    else:
        x0 = model.prepare_content(batch)["content_token"]
    cond_emb = dt._cond(cond.get("condition_token"), cond.get("condition_embed_token")).to(dev)
    return x0.contiguous(), cond_emb.contiguous()


@torch.no_grad()
def training_draws(model, x0, cond_emb, generator=None):
    """... and the part that depends on the training state: sample_time reads the importance-sampling statistics the previous
    iteration updated.  -> (x0, cond_emb, t, pt, noise)"""
    dt = model.transformer
    dev = dt.device
    B = x0.shape[0]
    t, pt = dt.sample_time(B, dev, "importance", generator=generator)
    noise = torch.rand((B, dt.num_classes, dt.content_seq_len), device=dev, generator=generator)
    return x0, cond_emb, t.to(dev), pt.to(dev), noise


class TrainStep:
    def __init__(self, diffusion_transformer, precision="fp32", rescale_interval=100, attention="fused"):
        assert precision in ("f16x2", "fp32") and attention in ("fused", "composed")
        self.attention = attention
        # fused attention FORWARD on the streamed fp16-split kernel of the sampling path (ds_attention_f16x2) in the "f16x2"
        # backend: 122 -> ~35 us per self-attention launch at B = 20; gradient parity unchanged (tests/test_hip_train_kernels.py)
        self.split_attention = precision == "f16x2"
        self.dt = diffusion_transformer
        self.tr = diffusion_transformer.transformer
        self.precision = precision
        self.gemm = _SplitGemm() if precision == "f16x2" else _Fp32Gemm()
        self.rescale_interval = rescale_interval
        self.loss_scale_exp = None if precision == "f16x2" else 0    # k of the loss scale 2^k; None: calibrate first
        self.calibrated_amax = None
        self._steps = 0
        self._capturing = False     # set by GraphedIteration while a hipGraph records the step: no host syncs then
        self._calib_norm = None     # global gradient norm at calibration time (observe_grad_norm)
        # Saturation monitor of the split backend (ds_split_hi / _lo SATURATE at 65504 -- no inf / NaN ever shows that a
        # gradient left the calibrated window): every step folds max |scaled dY| over all GEMM inputs into this device
        # scalar (ds_amax, the calibration's own probe; captured into the graph like any other launch), and
        # check_loss_scale() reads it on the host every `monitor_interval` steps -- whether or not clipping is configured.
        self.monitor_interval = 16
        # Where a calibration puts the largest value of every fp16-split gradient operand of ITS batch: 2^calib_log2 .. 2x that.
        # What matters for precision is only that a tensor's largest element is >= 2^0 (a split value keeps 22 bits down to
        # 2^-3 and 2^-25 absolutely under that: with the maximum at 2^T every element errs by <= 2^-(25+T) of it -- fp32's own
        # 2^-24 at T = 0), and since round 6 EVERY site has its own power of two (calibrate: `_site_exp`), so the target can sit
        # low and leave the room above to the batches: rounds 3-5 put ONE global maximum at 2^12, three bits under the window's
        # upper bound, and the measured batch-to-batch spread of that maximum is three bits (most batches 2^-8.7, every fifth
        # 2^-5.73 = (1 / pt) / (B L): one position whose d logit is ~1) -- a run re-captured as soon as its first large batch
        # came by (profiles/r05last_monitor_ab.txt); per site the spread is another 2.5 bits (profiles/r06b_*).  6 leaves nine.
        self.calib_log2 = 6
        self.monitor_window = (0, 15)           # log2 bounds of max |scaled operand| outside which the calibration is dropped
        self.monitor_log = []                   # log2 of the last readings (host floats; tools/bench_train.py prints them)
        # what a HIGH reading teaches: by how many bits later calibrations aim lower (a calibration looks at ONE batch; the
        # excursion that tripped the monitor is then put at 2^12).  Forgotten after `cap_decay_readings` quiet readings in a
        # row, when a reading falls under the window, and when the weights are replaced.
        self._target_drop = 0
        self._site_exp = None                   # {linear key: e}: the site's own 2^e on top of the loss scale (calibrate)
        self._site_order = []
        self._clean_readings = 0                # consecutive readings under 2^11 while a drop is in force
        self.cap_decay_readings = 32            # ... after that many (512 iterations) the drop is forgotten
        self._amax_live = None
        self._since_check = 0
        # A calibration has seen ONE batch: the monitor is read after 1, 2, 4, 8 iterations before it settles at every
        # `monitor_interval`-th -- on trained-like weights a batch 2^11 above the calibration batch came by within the first 16
        # iterations (profiles/r06k_bench_train_long_runs.txt: reading 2^17.65, i.e. saturated planes until the check)
        self._next_check = 1
        # weight swaps outside this class (checkpoint / EMA loads: solver._invalidate) must drop the cached pre-scales
        import weakref
        diffusion_transformer.__dict__.setdefault("_scale_clients", []).append(weakref.ref(self))

    def reset_scales(self, weights_replaced=True):
        """Forget the per-matrix weight pre-scales 2^s and the loss scale: the next step re-derives both (one calibration
        backward).  Called after the weights were replaced behind this object's back (solver._invalidate) -- then what the
        saturation monitor had learnt about the OLD weights' gradients (`_target_drop`) goes too -- and, with
        weights_replaced=False, by a re-capture of the same run (the bound is exactly what that re-calibration needs)."""
        if weights_replaced:
            self._target_drop, self._clean_readings = 0, 0
        if hasattr(self.gemm, "wexp"):
            self.gemm.wexp.clear()
        if self.precision == "f16x2":
            self.loss_scale_exp = None
            self._site_exp = None
        self._calib_norm = None
        if self._amax_live is not None:
            self._amax_live.zero_()
        # a captured iteration (GraphedIteration) has the OLD pre-scales and loss scale baked into its graph: it must be
        # re-captured BEFORE its next replay (swapped-in weights a few times larger would saturate the fp16 planes silently)
        self._scales_epoch = getattr(self, "_scales_epoch", 0) + 1

    def check_loss_scale(self, force=False):
        """Host side of the saturation monitor: every `monitor_interval` calls (after 1, 2, 4, 8 calls right behind a
        calibration; or when forced) read max |scaled operand| over all sites since the last check (one sync) and drop the
        calibration when it has left `monitor_window` = [2^0, 2^15) -- fp16 saturates at 2^16, and a site whose largest element
        is under 2^0 no longer has fp32-class planes.  Returns True when the next step must re-calibrate (a captured iteration
        must then be re-captured)."""
        if self.precision != "f16x2" or self._amax_live is None:
            return False
        self._since_check += 1
        if not force and self._since_check < min(self._next_check, self.monitor_interval):
            return False
        self._since_check = 0
        self._next_check = min(self.monitor_interval, 2 * self._next_check)
        m = float(self._amax_live.item())
        self._amax_live.zero_()
        self.monitor_log = self.monitor_log[-63:] + [round(math.log2(m), 2) if m > 0.0 and math.isfinite(m) else m]
        if m == 0.0:
            # ds_amax never lets a NaN win and skips non-positive values, so 0 means EITHER genuinely zero gradients (nothing
            # was scaled: no reason to re-calibrate / re-capture) OR an all-NaN scaled dY (a diverged loss).  The loss of the
            # same step tells them apart at this very host sync.
            last = getattr(self, "_last_loss", None)
            if last is not None and not math.isfinite(float(last)):
                # a diverged run: no loss scale repairs it, and answering True here would re-calibrate (and re-capture a graphed
                # iteration) at every monitor interval for the rest of the run
                raise FloatingPointError("training diverged: the loss is %r (every scaled gradient is NaN)" % float(last))
            return False
        lo, hi = self.monitor_window
        if math.isfinite(m) and 2.0 ** lo <= m < 2.0 ** hi:
            # inside the window.  What one excursion taught must not hold the target down for ever: once the readings have
            # stayed under 2^11 for `cap_decay_readings` checks in a row it is forgotten (the scales themselves are left
            # alone -- the next re-calibration, whenever something asks for one, aims at the full target again)
            if self._target_drop:
                self._clean_readings = self._clean_readings + 1 if m < 2.0 ** 11 else 0
                if self._clean_readings >= self.cap_decay_readings:
                    self._target_drop, self._clean_readings = 0, 0
            return False
        if math.isfinite(m) and m >= 2.0 ** hi:
            # put THIS excursion at 2^12 from now on: aim that many bits lower (never under 2^1)
            self._target_drop = min(self.calib_log2 - 1, self._target_drop + math.floor(math.log2(m)) - 12)
            self._clean_readings = 0
            self.last_trip = "monitor high: max |scaled operand| = 2^%.2f" % math.log2(m)
        elif math.isfinite(m):
            self._target_drop = 0                                              # gradients have shrunk: aim at the full target again
            self.last_trip = "monitor low: max |scaled operand| = 2^%.2f" % math.log2(m)
        else:
            self.last_trip = "monitor: max |scaled operand| = %r" % m
        self.loss_scale_exp, self._calib_norm = None, None
        return True

    def _target(self):
        """log2 of where calibrations put a site's largest operand value right now (calib_log2 minus what excursions taught)"""
        return max(1, self.calib_log2 - self._target_drop)

    def _exp_from_amax(self, m):
        """exponent k that puts a largest value m at 2^target .. 2^(target + 1)"""
        if m == 0.0 or not math.isfinite(m):
            return 0
        return self._target() - math.floor(math.log2(m))

    @torch.no_grad()
    def prescales_drifted(self):
        """Host check of the weight pre-scales 2^s a CAPTURED iteration froze (the eager step simply refreshes them every
        `rescale_interval` steps): one host sync over max |W| of every matrix.  True when a matrix has outgrown its place --
        max |W| 2^s has reached 2^14, one bit of fp16's range left for the replays until the next check -- or has fallen more
        than two bits under it (2 of the split's 22 bits lost), or when there are no pre-scales at all (weights were swapped)."""
        if self.precision != "f16x2":
            return False
        old = self.gemm.wexp
        if not old:
            return True
        blocks, lin_logits = self._linears()
        new = self.gemm.scales_of([l for b in blocks for l in b.values()] + [lin_logits])
        return any(k not in old or s < old[k] or s > old[k] + 2 for k, s in new.items())

    def observe_grad_norm(self, norm):
        """Second guard of the calibrated scales ("f16x2" backend, eager solver): the calibration leaves 2^9 of headroom below
        fp16's range and 2^6 above the point where the largest element of an operand would fall under 2^0.  Gradients grow and
        shrink together, so the global gradient norm the solver computes anyway is a monitor too: once it has moved by more
        than 32x up or 64x down from its value at calibration time, the next step re-calibrates (returns True then).  Call
        it with a HOST float (the solvers do, next to float(loss))."""
        if self.precision != "f16x2" or not math.isfinite(norm) or norm <= 0.0:
            return False
        if self._calib_norm is None:
            self._calib_norm = norm
            return False
        if norm > 32.0 * self._calib_norm or norm < self._calib_norm / 64.0:
            self.loss_scale_exp, self._calib_norm = None, None
            return True
        return False

    # ---- the step's linears ------------------------------------------------------------------------------------------
    def _linears(self):
        """[(per-block dict), ..., logits]: fused weights are concatenated here once per step (query | key | value rows)."""
        tr = self.tr
        out = []
        for li, blk in enumerate(tr.blocks):
            a1, a2 = blk.attn1, blk.attn2
            p = "b%d." % li
            out.append({
                "qkv1": _Linear(p + "qkv1", (a1.query.weight, a1.key.weight, a1.value.weight),
                                (a1.query.bias, a1.key.bias, a1.value.bias)),
                "proj1": _Linear(p + "proj1", a1.proj.weight.detach(), a1.proj.bias.detach()),
                "q2": _Linear(p + "q2", a2.query.weight.detach(), a2.query.bias.detach()),
                "kv2": _Linear(p + "kv2", (a2.key.weight, a2.value.weight), (a2.key.bias, a2.value.bias)),
                "proj2": _Linear(p + "proj2", a2.proj.weight.detach(), a2.proj.bias.detach()),
                "fc1": _Linear(p + "fc1", blk.mlp[0].weight.detach(), blk.mlp[0].bias.detach()),
                "fc2": _Linear(p + "fc2", blk.mlp[2].weight.detach(), blk.mlp[2].bias.detach()),
            })
        lin = tr.to_logits[1]
        return out, _Linear("logits", lin.weight.detach(), lin.bias.detach())

    @torch.no_grad()
    def loss_and_grads(self, x0, cond_emb, t, pt, noise, on_grads=None):
        """x0 i64[B, L] clean tokens, cond_emb f32[B, 77, 512], t i64[B], pt f32[B] (sample_time's output), noise
        f32[B, K+1, L] uniforms for q_sample.  Returns (loss scalar as forward() reports it, {parameter name relative to
        the DiffusionTransformer: gradient}).  Gradients are those of that loss.
        on_grads(named, streams): called during the backward -- after the logits layer and after every transformer block,
        last block first -- with the WEIGHT-matrix gradients that are final at that point (76 % of the bytes), then once
        after the block loop with the AdaLN tables' parameter gradients (22 %: they come out of two grouped GEMMs over all
        modules); biases and norm gains are un-scaled in one multiply at the very end and stay with finish().  `streams`:
        the streams that wrote them.  The hook of the overlapped data-parallel reduction (shard.GradientReducer.ready)."""
        if self.loss_scale_exp is None:
            self.calibrate(x0, cond_emb, t, pt, noise)
        return self._run(x0, cond_emb, t, pt, noise, calibrating=False, on_grads=on_grads)

    @torch.no_grad()
    def calibrate(self, x0, cond_emb, t, pt, noise):
        """Loss scale of the "f16x2" backend from one unscaled backward on this batch: the largest |dY| that enters a GEMM
        is put at 2^12..2^13 (fp16 overflows at 2^16; a split value keeps 2^-25 absolutely).  One host sync.  The
        importance-sampling statistics (Lt_history / Lt_count) are not touched."""
        if self.precision != "f16x2":
            return 0
        self.loss_scale_exp, self._site_exp = 0, None
        self._next_check, self._since_check = 1, 0
        amax = torch.zeros(1, device=x0.device)
        self._run(x0, cond_emb, t, pt, noise, calibrating=True, amax=amax)
        m = float(amax.item())
        self.calibrated_amax = m
        k0 = self._exp_from_amax(m)
        # Second pass, under that provisional scale (unscaled, the deep sites' operands flush to 0): the largest value of EVERY
        # operand the backward splits to fp16 under the loss scale -- max |dY| per linear (its dY feeds the dX and dW GEMMs)
        # and max |dO| over the attention backwards -- in one more host sync.  One scale for the whole backward leaves the small gradients
        # behind: against the reference at 19 layers / B = 20 (tests/test_hip_train_batch.py) the cross-attention query
        # projections -- whose dY is a softmax gradient of near-uniform probabilities, 2^-14 of the largest dY -- came out with
        # 1e-2 relative error, their fp16 lo plane under the subnormal range (the reference's own fp32: 6e-7).  So
        # every SITE gets its own power of two on top of the loss scale (the fp32 tensors in between carry the loss scale without
        # harm, whatever it is): a linear's dY 2^e is what is split (ds_pack_operand `scale`) and 2^-e goes into its dX / dW
        # epilogues; an attention backward's dO 2^e is what its kernels split and their stores take 2^-e out again (they
        # normalise their in-register dS by themselves) -- all exact.
        self.loss_scale_exp = k0
        n_sites = 9 * len(self.tr.blocks) + 1                                # 7 linears + 2 attentions per block, + the logits layer
        sites = torch.zeros(n_sites, device=x0.device)                       # in the order the backward visits them
        self._run(x0, cond_emb, t, pt, noise, calibrating=True, amax=torch.zeros(1, device=x0.device), site_amax=sites)
        per_site = sites.tolist()
        assert len(self._site_order) == n_sites or self.attention != "fused", (len(self._site_order), n_sites)
        self._site_exp = {k: (0 if (v == 0.0 or not math.isfinite(v)) else
                              max(-40, min(40, self._target() - math.floor(math.log2(v)))))
                          for k, v in zip(self._site_order, per_site)}
        return self.loss_scale_exp

    def _run(self, x0, cond_emb, t, pt, noise, calibrating, amax=None, on_grads=None, site_amax=None):
        """site_amax: second calibration pass -- a zeroed f32[>= number of linears] whose slot i takes max |dY| of the i-th
        linear's output gradient IN THE ORDER THE BACKWARD VISITS THEM (self._site_order receives the keys)"""
        dt, tr, G_ = self.dt, self.tr, self.gemm
        site_exp = {} if (calibrating or self._site_exp is None) else self._site_exp
        site_index = {}
        dev = x0.device
        B, Lx = x0.shape
        D, H, K = tr.n_embd, tr.n_head, tr.num_codes
        M = B * Lx
        T = dt.num_timesteps
        G_.rows_per_sample = Lx
        blocks, lin_logits = self._linears()
        all_lins = [l for b in blocks for l in b.values()] + [lin_logits]
        if self.precision == "f16x2" and (not G_.wexp or (self._steps % self.rescale_interval == 0 and not calibrating
                                                           and not self._capturing)):
            G_.refresh_scales(all_lins)
        for l in all_lins:
            G_.prepare(l)
        scale = 2.0 ** self.loss_scale_exp
        inv = 1.0 / scale
        fused = self.attention == "fused"

        if amax is None and self.precision == "f16x2":  # the saturation monitor (check_loss_scale)
            if self._amax_live is None or self._amax_live.device != dev:
                self._amax_live = torch.zeros(1, device=dev)
            amax = self._amax_live

        # (every gradient that enters a GEMM passes G_.prep_dy, whose pack folds max |dY 2^e| into `amax`: calibration / monitor)
        def att_site(key):      # (monitor slot, this attention backward's own power of two)
            if site_amax is not None:
                return site_amax[site_index.setdefault(key, len(site_index))], 1.0
            return amax, 2.0 ** site_exp.get(key, 0)

        sched = dt._schedule_table()
        xt = dt.q_sample_tokens(x0.contiguous(), t, noise)
        emb = tr.content_emb
        pos = emb.position_table()
        x = torch.empty(M, D, device=dev)
        L_.check(L_.lib().ds_embed(L_.ptr(xt), L_.ptr(emb.emb.weight), L_.ptr(pos), L_.ptr(x), M, Lx, D, L_.stream()))
        cond = cond_emb.reshape(-1, cond_emb.shape[-1]).float().contiguous()
        Lc = cond_emb.shape[1]
        # operand handles (G_.prep_x): the forward's GEMM input AND the X^T of the same layer's dW, made in one pass; the
        # caption embedding feeds every block's cross K | V projection (same K, same padded contraction: one handle)
        cond_h = G_.prep_x(blocks[0]["kv2"], cond)
        # the 2 n_layer AdaLN modulations  Linear(SiLU(Emb(t_b)))  (transformer_utils.py:145-147) as ONE grouped exact-fp32 GEMM over
        # the batch's OWN timesteps -- B rows per module, as the reference computes them (the sampling loop tabulates all T rows
        # once; here the weights change every iteration and a table of 100 rows was 5x the work: 416 -> 90 us, and as much again
        # in the backward).  The AdaLN kernels index the [B][2D] rows with sample_rows = 0 .. B-1.
        lns = [ln for blk in tr.blocks for ln in (blk.ln1, blk.ln1_1)]
        ada_E = torch.stack([ln.emb.weight.detach() for ln in lns]).index_select(1, t)   # [G][B][D] = Emb(t_b)
        ada_W = torch.stack([ln.linear.weight.detach() for ln in lns])                 # [G][2D][D]
        sample_rows = torch.arange(B, device=dev)
        tabs = torch.empty(len(lns), B, 2 * D, device=dev)
        L_.gemm(torch.nn.functional.silu(ada_E), ada_W, tabs, B, 2 * D, D, groups=len(lns), a_gstride=B * D,
                w_gstride=2 * D * D, c_gstride=B * 2 * D)
        tabs += torch.stack([ln.linear.bias.detach() for ln in lns])[:, None, :]
        saved = []
        for li, (blk, ls) in enumerate(zip(tr.blocks, blocks)):
            s = {"x0": x}
            s["tab1"] = tabs[2 * li]
            s["h1"] = h = G_.prep_x(ls["qkv1"], _norm_fwd(x, 0, Lx, table=s["tab1"], t=sample_rows))
            qkv = G_.fwd(ls["qkv1"], h)                                             # [M][3D]: q | k | v
            if fused:
                s["att1"] = _FusedAttn((qkv, 0, 3 * D), (qkv, D, 3 * D), (qkv, 2 * D, 3 * D), B, Lx, Lx, H, split=self.split_attention)
            else:
                s["att1"] = _Attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, Lx, Lx, H)
            s["o1"] = G_.prep_x(ls["proj1"], s["att1"].out)
            x = G_.fwd(ls["proj1"], s["o1"], R=x)
            s["x1"] = x
            s["tab2"] = tabs[2 * li + 1]
            s["h2"] = h = G_.prep_x(ls["q2"], _norm_fwd(x, 0, Lx, table=s["tab2"], t=sample_rows))
            q = G_.fwd(ls["q2"], h)
            kv = G_.fwd(ls["kv2"], cond_h)                                          # [B*Lc][2D]: k | v
            if fused:
                s["att2"] = _FusedAttn((q, 0, D), (kv, 0, 2 * D), (kv, D, 2 * D), B, Lx, Lc, H, split=self.split_attention)
            else:
                s["att2"] = _Attn(q, kv[:, :D], kv[:, D:], B, Lx, Lc, H)
            s["o2"] = G_.prep_x(ls["proj2"], s["att2"].out)
            x = G_.fwd(ls["proj2"], s["o2"], R=x)
            s["x2"] = x
            s["h3"] = h = G_.prep_x(ls["fc1"], _norm_fwd(x, 1, Lx, gamma=blk.ln2.weight, beta=blk.ln2.bias))
            u = G_.fwd(ls["fc1"], h)
            s["u"] = u
            s["g"] = G_.prep_x(ls["fc2"], u, pro=PACK_GELU2)                        # gelu2(u): the f16x2 backend never stores it in fp32
            x = G_.fwd(ls["fc2"], s["g"], R=x)
            saved.append(s)
        xf = x
        lnf = tr.to_logits[0]
        hf = G_.prep_x(lin_logits, _norm_fwd(xf, 1, Lx, gamma=lnf.weight, beta=lnf.bias))
        logits = G_.fwd(lin_logits, hf)                                             # [M, K]
        # ---- loss (forward value) and d loss / d logits
        kl, nll, kl_aux = (torch.empty(B, Lx, device=dev) for _ in range(3))
        L_.check(L_.lib().ds_loss_tail(L_.ptr(logits), L_.ptr(x0), L_.ptr(xt), L_.ptr(t), L_.ptr(sched), L_.ptr(kl), L_.ptr(nll),
                                       L_.ptr(kl_aux), None, B, Lx, K, T, L_.stream()))
        mask_region = (xt == K).float()
        weight = mask_region * dt.mask_weight[0] + (1.0 - mask_region) * dt.mask_weight[1]
        is0 = (t == 0).float()
        kl_loss = is0 * nll.sum(-1) + (1.0 - is0) * (kl * weight).sum(-1)
        if not calibrating:
            lt2 = kl_loss.pow(2)                     # importance-sampling statistics of sample_time (:452-455)
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
        dlog.mul_(norm * scale)                                                      # loss = sum(vb) / (B L); x loss scale
        # ---- backward (every d* below carries the loss scale; `small` collects what one multiply un-scales at the end)
        g, small = {}, []

        def lin_bwd(lin, xh, dy, need_dx=True, pro=PACK_PLAIN, aux=None):
            """The three products of one linear layer for its output gradient dy (fp32; with pro = PACK_GELU2_BWD the
            gradient of the layer's output is dy * gelu2'(aux)).  dy is packed ONCE (G_.prep_dy: row form, transposed form,
            bias column sums, max |dY|), then dX, dW and db.  (Rounds 2-4 could put dW / db on a second HIP stream; measured
            slower in both rounds it was tried -- 15.6 vs 15.9 and 16.2 vs 16.6 it/s, the GEMMs fill the power-capped chip -- and
            removed in round 5.)"""
            # the site's own power of two on top of the loss scale (module docstring; 1 while calibrating)
            e = site_exp.get(lin.key, 0)
            up, down = 2.0 ** e, 2.0 ** -e
            site_slot = amax if site_amax is None else site_amax[site_index.setdefault(lin.key, len(site_index))]
            dyh = G_.prep_dy(lin, dy, pro=pro, aux=aux, amax=site_slot, need_row=need_dx, scale=up)
            dxo = G_.dx(lin, dyh, unscale=down) if need_dx else None
            if hasattr(G_, "dw_many"):           # off the critical path: collected, launched side by side (flush_dw)
                dW = torch.empty(lin.N, lin.K, device=dev)
                pending_dw.append((lin, xh, dyh, inv * down, dW))
            else:
                dW = G_.dw(lin, xh, dyh, inv * down)
            db = G_.db(lin, dyh)
            small.append(db)
            return dxo, dW, db

        pending_dw, pending_names = [], []

        def flush_dw():
            if pending_dw:
                G_.dw_many(pending_dw)
                pending_dw.clear()

        def hand_over(names, now=True):
            """now=False: the names wait until the weight gradients of TWO blocks are collected (their products then pair up:
            two qkv gradients in one grid, four MLP ones, ...) -- the overlapped reduction gets them one block later"""
            pending_names.extend(names)
            if not now and len(pending_dw) < 14:
                return
            flush_dw()                           # (the gradients handed over must be final)
            if on_grads is not None and not calibrating:
                on_grads({n: g[n] for n in pending_names}, ())
            pending_names.clear()

        dh, g["transformer.to_logits.1.weight"], g["transformer.to_logits.1.bias"] = lin_bwd(lin_logits, hf, dlog)
        hand_over(["transformer.to_logits.1.weight"])
        dx, dgam, dbet = _norm_bwd(xf, dh, 1, Lx, gamma=lnf.weight)
        g["transformer.to_logits.0.weight"], g["transformer.to_logits.0.bias"] = dgam[0], dbet[0]
        small += [dgam, dbet]

        ada = []                    # (AdaLayerNorm module, d scale [B][D], d shift [B][D], parameter prefix): batched below

        def adaln_param_grads(ln, d_scale, d_shift, pfx):
            # (d_scale / d_shift are the two column halves of _norm_bwd's [B][2D] sums: ._base is [d scale | d shift] itself)
            both = d_scale._base if d_scale._base is not None and d_scale._base is d_shift._base else torch.cat((d_scale, d_shift), dim=1)
            ada.append((ln, both, None, pfx))

        def adaln_param_grads_all():
            """d modulation rows [B][2D] -> emb.weight / linear.{weight, bias} through  mod_b = Linear(SiLU(emb[t_b]))  (AdaLayerNorm,
            transformer_utils.py:134-149) for ALL 2 n_layer AdaLN modules at once: dW = dmod^T silu(e_b) (ds_rows_outer), db = column
            sums of dmod, de[t_b] += (dmod_b W) silu'(e_b) (ds_rows_times_matrix): two passes over the B samples for all modules
            instead of two small GEMMs, three transposing copies and a dozen elementwise launches per module (4 ms of an 82 ms
            iteration in round 4; grouped GEMMs over all T table rows until round 6)."""
            G = len(ada)
            if G == 0:
                return
            ada.sort(key=lambda a: lns.index(a[0]))             # forward order (the backward visited the blocks last first):
            order = [lns.index(ln) for ln, _, _, _ in ada]      # the stacked parameters of the forward are used as they are
            dmod = torch.stack([both for _, both, _, _ in ada]) * inv                               # [G][B][2D]
            E = ada_E if order == list(range(len(lns))) else torch.stack([ada_E[i] for i in order])   # [G][B][D]
            sg = torch.sigmoid(E)
            dw = torch.empty(G, 2 * D, D, device=dev)                                              # dW[g] = dmod[g]^T silu(e_b[g]):
            silu_e = (E * sg).contiguous()                                                         # B outer products per module
            L_.check(L_.lib().ds_rows_outer(L_.ptr(dmod), L_.ptr(silu_e), L_.ptr(dw), G, B, 2 * D, D, L_.stream()))
            # dmod[g] W[g]: B rows against the row-major weights where they lie (ds_rows_times_matrix: no transposed copy of the
            # 38 matrices -- 0.38 ms per iteration -- and no tile program that is mostly padding rows); ada_W is in forward order
            Wg = ada_W if order == list(range(len(lns))) else torch.stack([ada_W[i] for i in order])
            KS = (2 * D) // 256
            part = torch.empty(KS, G, B, D, device=dev)
            L_.check(L_.lib().ds_rows_times_matrix(L_.ptr(dmod), L_.ptr(Wg), L_.ptr(part), G, B, 2 * D, D, L_.stream()))
            ds_ = torch.empty(G, B, D, device=dev)
            L_.check(L_.lib().ds_colsum(L_.ptr(part), L_.ptr(ds_), 1, KS, G * B * D, G * B * D, 0, 0, L_.stream()))
            de = torch.zeros(G, T, D, device=dev)
            de.index_add_(1, t, ds_ * (sg * (1.0 + E * (1.0 - sg))))                               # samples that share a timestep add up
            dbias = dmod.sum(1)                                                                    # [G][2D]
            for i, (_, _, _, pfx) in enumerate(ada):
                g[pfx + ".emb.weight"], g[pfx + ".linear.weight"], g[pfx + ".linear.bias"] = de[i], dw[i], dbias[i]

        for li in reversed(range(len(saved))):
            s, blk, ls = saved[li], tr.blocks[li], blocks[li]
            p = "transformer.blocks.%d." % li
            # x3 = x2 + fc2(gelu(fc1(ln2(x2))))
            dgact, g[p + "mlp.2.weight"], g[p + "mlp.2.bias"] = lin_bwd(ls["fc2"], s["g"], dx)
            # d fc1-output = dgact * gelu2'(u): the prologue of fc1's gradient pack
            dh, g[p + "mlp.0.weight"], g[p + "mlp.0.bias"] = lin_bwd(ls["fc1"], s["h3"], dgact, pro=PACK_GELU2_BWD, aux=s["u"])
            _, dgam, dbet = _norm_bwd(s["x2"], dh, 1, Lx, gamma=blk.ln2.weight, add_to=dx)
            g[p + "ln2.weight"], g[p + "ln2.bias"] = dgam[0], dbet[0]
            small += [dgam, dbet]
            # x2 = x1 + proj2(attn2(q(ln1_1(x1)), kv(cond)))
            dao, g[p + "attn2.proj.weight"], g[p + "attn2.proj.bias"] = lin_bwd(ls["proj2"], s["o2"], dx)
            dq = torch.empty(M, D, device=dev)
            dkv = torch.empty(B * Lc, 2 * D, device=dev)
            if fused:
                slot, dsc = att_site("b%d.att2" % li)
                s["att2"].backward(dao, (dq, 0, D), (dkv, 0, 2 * D), (dkv, D, 2 * D), amax=slot, do_scale=dsc)
            else:
                s["att2"].backward(dao, dq, dkv[:, :D], dkv[:, D:])
            dh, g[p + "attn2.query.weight"], g[p + "attn2.query.bias"] = lin_bwd(ls["q2"], s["h2"], dq)
            _, dWkv, dbkv = lin_bwd(ls["kv2"], cond_h, dkv, need_dx=False)
            g[p + "attn2.key.weight"], g[p + "attn2.value.weight"] = dWkv[:D], dWkv[D:]
            g[p + "attn2.key.bias"], g[p + "attn2.value.bias"] = dbkv[:D], dbkv[D:]
            _, dsc, dsh = _norm_bwd(s["x1"], dh, 0, Lx, table=s["tab2"], t=sample_rows, add_to=dx)
            adaln_param_grads(blk.ln1_1, dsc, dsh, p + "ln1_1")
            # x1 = x0 + proj1(attn1(qkv(ln1(x0))))
            dao, g[p + "attn1.proj.weight"], g[p + "attn1.proj.bias"] = lin_bwd(ls["proj1"], s["o1"], dx)
            dqkv = torch.empty(M, 3 * D, device=dev)
            if fused:
                slot, dsc = att_site("b%d.att1" % li)
                s["att1"].backward(dao, (dqkv, 0, 3 * D), (dqkv, D, 3 * D), (dqkv, 2 * D, 3 * D), amax=slot, do_scale=dsc)
            else:
                s["att1"].backward(dao, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
            dh, dWqkv, dbqkv = lin_bwd(ls["qkv1"], s["h1"], dqkv)
            for j, nm in enumerate(("query", "key", "value")):
                g[p + "attn1.%s.weight" % nm], g[p + "attn1.%s.bias" % nm] = dWqkv[j * D:(j + 1) * D], dbqkv[j * D:(j + 1) * D]
            _, dsc, dsh = _norm_bwd(s["x0"], dh, 0, Lx, table=s["tab1"], t=sample_rows, add_to=dx)
            adaln_param_grads(blk.ln1, dsc, dsh, p + "ln1")
            hand_over([p + n for n in ("mlp.2.weight", "mlp.0.weight", "attn2.proj.weight", "attn2.query.weight",
                                       "attn2.key.weight", "attn2.value.weight", "attn1.proj.weight", "attn1.query.weight",
                                       "attn1.key.weight", "attn1.value.weight")], now=False)
        hand_over([])                            # the blocks still waiting
        adaln_param_grads_all()
        # the AdaLN parameter gradients (17 MB per block, 22 % of the gradient bytes) are final here: hand them to the
        # overlapped reduction before the embedding backward and the closing un-scale instead of leaving them to finish()
        hand_over([pfx + sfx for _, _, _, pfx in ada for sfx in (".linear.weight", ".emb.weight")])
        # ---- embedding
        demb = torch.zeros_like(emb.emb.weight)
        L_.check(L_.lib().ds_embed_bwd(L_.ptr(dx), L_.ptr(xt), L_.ptr(demb), M, D, demb.shape[0], L_.stream()))
        g["transformer.content_emb.emb.weight"] = demb
        dpos = torch.empty(Lx, D, device=dev)
        L_.check(L_.lib().ds_colsum(L_.ptr(dx), L_.ptr(dpos), Lx, B, D, Lx * D, D, 0, L_.stream()))
        Hh, Ww = emb.spatial_size
        dpos3 = dpos.view(Hh, Ww, D)
        dhh = _colsum(dpos3.reshape(Hh * Ww, D), Hh)                                                    # sum over w
        g["transformer.content_emb.height_emb.weight"] = dhh
        dw = torch.empty(Ww, D, device=dev)
        L_.check(L_.lib().ds_colsum(L_.ptr(dpos), L_.ptr(dw), Ww, Hh, D, Ww * D, D, 0, L_.stream()))   # sum over h
        g["transformer.content_emb.width_emb.weight"] = dw
        small += [demb, dhh, dw]
        if inv != 1.0:
            torch._foreach_mul_(small, inv)
        if site_amax is not None:
            self._site_order = list(site_index)
        if not calibrating:
            self._steps += 1
            self._last_loss = loss          # (device scalar; a captured iteration keeps updating this very tensor)
        return loss, g

    # ---- optimizer -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def adamw_step(self, grads, state, step, lr, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2, hyper=None):
        """In-place AdamW on the parameters that have a gradient (state: dict name -> (m, v), created on first use).
        hyper: optional f32[4] device tensor { lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale } -- then `step` / `lr`
        are ignored and the launch arguments do not depend on the iteration (captured graphs: `capture`)."""
        params = dict(self.dt.named_parameters())
        if hyper is not None:
            # one batched launch per 64 tensors (ds_adamw_multi: descriptors by value in the kernel arguments -- capturable)
            import ctypes
            names = list(grads)
            rec = (ctypes.c_int64 * (5 * len(names)))()
            keep = []
            for i, name in enumerate(names):
                p_ = params[name]
                if name not in state:
                    state[name] = (torch.zeros_like(p_), torch.zeros_like(p_))
                m, v = state[name]
                gr = grads[name].contiguous()
                keep.append(gr)
                assert gr.numel() == p_.numel() and p_.data.is_contiguous() and m.is_contiguous() and v.is_contiguous()
                rec[5 * i:5 * i + 5] = [p_.data.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), p_.numel()]
            L_.check(L_.lib().ds_adamw_multi(ctypes.cast(rec, ctypes.c_void_p), len(names), L_.ptr(hyper), betas[0], betas[1], eps,
                                             weight_decay, L_.stream()))
            self.tr.invalidate()
            return
        for name, gr in grads.items():
            p_ = params[name]
            if name not in state:
                state[name] = (torch.zeros_like(p_), torch.zeros_like(p_))
            m, v = state[name]
            gr = gr.contiguous()
            L_.check(L_.lib().ds_adamw(L_.ptr(p_.data), L_.ptr(gr), L_.ptr(m), L_.ptr(v), p_.numel(), lr, betas[0], betas[1],
                                       eps, weight_decay, step, L_.stream()))
        self.tr.invalidate()       # cached weight packs / AdaLN tables are stale now (frees the native handle too)

    # ---- the whole iteration as ONE hipGraph -----------------------------------------------------------------------------
    def capture(self, x0, cond_emb, t, pt, noise, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2, max_norm=None,
                reduce=None):
        """Capture  loss_and_grads -> global-norm clip -> AdamW  (engine/solver_spec.py:308-331's order) on static copies of
        the batch tensors into one hipGraph and return a `GraphedIteration`: `it(x0, cond_emb, t, pt, noise, lr)` copies
        the batch in, refreshes the 4 device scalars of the update and replays -- one graph launch per training iteration.
        Data-parallel form: `reduce(grads)` (e.g. shard.allreduce_gradients: averages the gradient tensors over the
        ranks, in place, on the current stream) makes it TWO graphs per rank -- gradients | clip + AdamW -- with the
        reduction enqueued between the two replays (engine/solver_spec.py:109: DDP reduces before the optimizer step).
        The loss scale and the weight pre-scales are the calibrated constants of capture time; `recapture()` refreshes them."""
        return GraphedIteration(self, (x0, cond_emb, t, pt, noise), betas, eps, weight_decay, max_norm, reduce=reduce)


class GraphedIteration:
    def __init__(self, step, batch, betas, eps, weight_decay, max_norm, reduce=None):
        self.step, self.betas, self.eps, self.weight_decay, self.max_norm = step, betas, eps, weight_decay, max_norm
        self.reduce = reduce                 # None: one graph; callable(grads): gradients graph | reduce | update graph
        self.static = [b.clone() for b in batch]
        self.hyper = torch.zeros(4, device=batch[0].device)
        self.opt_state = {}
        self.iteration = 0
        self.graph = None
        self.update_graph = None
        self._capture()

    @torch.no_grad()
    def _capture(self):
        st = self.step
        dev = self.static[0].device
        if st.loss_scale_exp is None:
            st.calibrate(*self.static)
        # warm-up on a side stream (lazy initialisation inside the library, allocator pools), with lr = 0 and every
        # statistic restored afterwards so that the warm-up leaves no trace in the training state
        dt = st.dt
        keep = (dt.Lt_history.clone(), dt.Lt_count.clone(), st._steps)
        self.hyper.copy_(torch.tensor([0.0, 1.0, 1.0, 0.0]))
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for name, (m, v) in self.opt_state.items():
            m.zero_()
            v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        self.update_graph = None
        self._scales_epoch = getattr(st, "_scales_epoch", 0)     # the scales this graph bakes in (TrainStep.reset_scales bumps it)
        st._capturing = True
        try:
            if self.reduce is None:
                with torch.cuda.graph(self.graph):
                    self._body()
            else:       # two segments sharing one memory pool: the gradient tensors of the first are the second's inputs
                with torch.cuda.graph(self.graph):
                    self._body_grads()
                self.update_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.update_graph, pool=self.graph.pool()):
                    self._body_update()
        finally:
            st._capturing = False
        dt.Lt_history.copy_(keep[0])
        dt.Lt_count.copy_(keep[1])
        st._steps = keep[2]

    def _body(self):                       # (the warm-up runs it un-reduced on every rank alike: lr = 0, state restored)
        self._body_grads()
        self._body_update()

    def _body_grads(self):
        self.loss, self.grads = self.step._run(*self.static, calibrating=False)

    def _body_update(self):
        st, loss, grads = self.step, self.loss, self.grads
        # global gradient norm + clip coefficient (torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), <= 1) in two
        # launches, the coefficient written straight into the AdamW kernel's hyper[3] (ds_grad_norm_multi)
        import ctypes
        tensors = [g_.contiguous() for g_ in grads.values()]
        assert all(g_.dtype == torch.float32 for g_ in tensors)
        rec = (ctypes.c_int64 * (2 * len(tensors)))()
        chunks = 0
        for i, g_ in enumerate(tensors):
            rec[2 * i:2 * i + 2] = [g_.data_ptr(), g_.numel()]
            chunks += (g_.numel() + 4095) // 4096
        part = torch.empty(chunks, dtype=torch.float64, device=self.hyper.device)
        total1 = torch.empty(1, device=self.hyper.device)
        L_.check(L_.lib().ds_grad_norm_multi(ctypes.cast(rec, ctypes.c_void_p), len(tensors), L_.ptr(part), chunks,
                                             float(self.max_norm) if self.max_norm is not None else 0.0, L_.ptr(total1),
                                             L_.ptr_off(self.hyper, 3) if self.max_norm is not None else None, L_.stream()))
        self._norm_keep = (tensors, part)              # alive until the launches are enqueued / for the life of the captured graph
        total = total1.reshape(())
        st.adamw_step(grads, self.opt_state, 0, 0.0, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                      hyper=self.hyper)
        self.loss, self.grad_norm, self.grads = loss, total, grads

    def recapture(self, static=None, reason="requested"):
        """New calibration of the loss scale / weight pre-scales on the current static batch (or `static`, the batch about
        to run), then a new graph.  `reason` is kept (recapture_reasons: what a long run's log shows next to its rate)."""
        if static is not None:
            for dst, src in zip(self.static, static):
                dst.copy_(src)
        self.recaptures = getattr(self, "recaptures", 0) + 1
        self.recapture_reasons = getattr(self, "recapture_reasons", []) + ["iteration %d: %s" % (self.iteration, reason)]
        self.step.reset_scales(weights_replaced=False)
        keep_state = {k: (m.clone(), v.clone()) for k, (m, v) in self.opt_state.items()}
        self._capture()
        for k, (m, v) in keep_state.items():
            self.opt_state[k][0].copy_(m)
            self.opt_state[k][1].copy_(v)
        self._replays = 0

    def state_dict(self):
        """What a resumed run needs of the captured iteration: the AdamW moments (they live in the graph's static tensors)
        and the bias-correction counter."""
        return {"iteration": self.iteration, "optimizer": {k: (m.clone(), v.clone()) for k, (m, v) in self.opt_state.items()}}

    @torch.no_grad()
    def load_state_dict(self, state):
        """In place into the tensors the graph updates (no re-capture needed)."""
        self.iteration = int(state["iteration"])
        missing = [k for k in self.opt_state if k not in state["optimizer"]]
        extra = [k for k in state["optimizer"] if k not in self.opt_state]
        if missing or extra:
            raise RuntimeError("GraphedIteration.load_state_dict: missing %s, unexpected %s" % (missing, extra))
        for k, (m, v) in state["optimizer"].items():
            self.opt_state[k][0].copy_(m)
            self.opt_state[k][1].copy_(v)

    @torch.no_grad()
    def __call__(self, x0, cond_emb, t, pt, noise, lr):
        if getattr(self.step, "_scales_epoch", 0) != self._scales_epoch:
            # the weights were replaced behind the graph (checkpoint / EMA load -> TrainStep.reset_scales): its frozen
            # pre-scales belong to the old weights.  Re-capture on THIS batch before anything is replayed.
            self.recapture(static=(x0, cond_emb, t, pt, noise), reason="weights replaced (reset_scales)")
        for dst, src in zip(self.static, (x0, cond_emb, t, pt, noise)):
            dst.copy_(src)
        self.iteration += 1
        b1, b2 = self.betas
        self.hyper[:3].copy_(torch.tensor([lr, 1.0 - b1 ** self.iteration, math.sqrt(1.0 - b2 ** self.iteration)]))
        if self.max_norm is None:
            self.hyper[3:4].fill_(1.0)
        self.graph.replay()
        if self.update_graph is not None:
            self.reduce(self.grads)        # in place on the graph's own gradient tensors, stream-ordered between the replays
            self.update_graph.replay()
        self.step._steps += 1
        self._replays = getattr(self, "_replays", 0) + 1
        self.step.tr.invalidate()          # the replay updated the weights: cached inference packs are stale
        return {"loss": self.loss, "grad_norm": self.grad_norm, "lr": lr}

    def check_loss_scale(self):
        """Host-side guards of the frozen constants of a captured iteration.  (1) The weight pre-scales 2^s are those of
        capture time -- the eager step refreshes them every `rescale_interval` steps, a replay cannot: after that many
        replays the weights are looked at (TrainStep.prescales_drifted, one host sync) and the iteration is re-captured only
        if a matrix has left its place.  (Until round 5 this re-captured unconditionally: ~1.9 s per 100 replays of 57 ms,
        a quarter of a long run -- profiles/r05last_monitor_ab.txt.)  (2) The saturation monitor (max |scaled dY|,
        TrainStep.check_loss_scale) is read every `monitor_interval` replays -- one host sync then, none in between -- and
        re-captures when the gradients have left the calibrated window."""
        st = self.step
        if getattr(self, "_replays", 0) >= max(1, st.rescale_interval):
            self._replays = 0
            if st.prescales_drifted():
                self.recapture(reason="a weight pre-scale 2^s left its place")
                return True
        if st.check_loss_scale():
            self.recapture(reason=getattr(st, "last_trip", "saturation monitor"))
            return True
        return False
