"""Row / elementwise kernels of the training step (csrc/train.hip, scope row 8f-3) against torch autograd computed on
the CPU in float64.  GPU only."""
import math

import pytest
import torch

from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu


def rnd(shape, key, scale=1.0):
    return (synth.synth_uniform(shape, key=key) * 2 - 1) * scale


@pytest.fixture(scope="module")
def L():
    from text_to_sound_synthesis_amd import _lib
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    _lib.lib()
    return _lib


def close(a, b, tol=2e-5):
    return (a.double() - b.double()).abs().max().item() <= tol * max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize("mode", [0, 1])
def test_layernorm_backward(L, mode):
    M, Lr, D, T = 530, 265, 1024, 100
    x = rnd((M, D), "lnb.x", 3.0).double().requires_grad_(True)
    dy = rnd((M, D), "lnb.dy")
    tab = rnd((T, 2 * D), "lnb.tab").double().requires_grad_(True)
    gamma = (rnd((D,), "lnb.g") + 1.5).double().requires_grad_(True)
    beta = rnd((D,), "lnb.b").double().requires_grad_(True)
    t = torch.tensor([3, 97])
    xn = torch.nn.functional.layer_norm(x, (D,), eps=1e-5)
    if mode == 0:
        e = tab[t]                                                      # [B, 2D]
        y = xn.view(2, Lr, D) * (1 + e[:, None, :D]) + e[:, None, D:]
        y = y.reshape(M, D)
    else:
        y = xn * gamma + beta
    y.backward(dy.double())
    xc, dyc = x.detach().float().cuda(), dy.cuda()
    dx = torch.full((M, D), float("nan"), device="cuda")
    dyxn = torch.full((M, D), float("nan"), device="cuda")
    tabc, tc, gc = tab.detach().float().cuda(), t.cuda(), gamma.detach().float().cuda()
    L.check(L.lib().ds_layernorm_bwd(L.ptr(xc), L.ptr(dyc), L.ptr(dx), L.ptr(dyxn), M, Lr, D, mode, L.ptr(tabc), L.ptr(tc),
                                     L.ptr(gc), L.stream()))
    assert close(dx.cpu(), x.grad)
    want = rnd((M, D), "lnb.res", 2.0).cuda() + dx                    # what accumulating into a residual gradient must give
    # the form with the scale / shift sums folded in (no dyxn matrix): same dx, sums per sample (mode 0) / over all rows (mode 1)
    G = 2 if mode == 0 else 1
    chunks = L.lib().ds_layernorm_bwd_chunks(M, Lr, mode)
    assert chunks == ((Lr if mode == 0 else M) + 15) // 16
    part = torch.full((G * chunks, 2 * D), float("nan"), device="cuda")
    dx2 = torch.full((M, D), float("nan"), device="cuda")
    L.check(L.lib().ds_layernorm_bwd_sums(L.ptr(xc), L.ptr(dyc), L.ptr(dx2), L.ptr(part), M, Lr, D, mode, L.ptr(tabc), L.ptr(tc),
                                          L.ptr(gc), 0, L.stream()))
    assert close(dx2.cpu(), dx.cpu(), 2e-6)            # (the same expressions; hipcc contracts the two kernels' multiply-adds differently)
    sums = torch.empty(G, 2 * D, device="cuda")
    L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(sums), G, chunks, 2 * D, 2 * D, chunks * 2 * D, 0, L.stream()))
    if mode == 0:
        assert close(sums.cpu()[:, :D], tab.grad[t][:, :D]) and close(sums.cpu()[:, D:], tab.grad[t][:, D:])
    else:
        assert close(sums.cpu()[0, :D], gamma.grad) and close(sums.cpu()[0, D:], beta.grad)
    res2 = rnd((M, D), "lnb.res", 2.0).cuda()
    L.check(L.lib().ds_layernorm_bwd_sums(L.ptr(xc), L.ptr(dyc), L.ptr(res2), L.ptr(part), M, Lr, D, mode, L.ptr(tabc), L.ptr(tc),
                                          L.ptr(gc), 1, L.stream()))
    assert close(res2.cpu(), want.cpu(), 2e-6)
    if mode == 0:   # per-sample scale / shift gradients = the rows of d tab[t]
        ds_ = torch.empty(2, D, device="cuda")
        db_ = torch.empty(2, D, device="cuda")
        L.check(L.lib().ds_colsum(L.ptr(dyxn), L.ptr(ds_), 2, Lr, D, D, Lr * D, 0, L.stream()))
        L.check(L.lib().ds_colsum(L.ptr(dyc), L.ptr(db_), 2, Lr, D, D, Lr * D, 0, L.stream()))
        assert close(ds_.cpu(), tab.grad[t][:, :D]) and close(db_.cpu(), tab.grad[t][:, D:])
    else:
        dg = torch.zeros(1, D, device="cuda")
        db_ = torch.zeros(1, D, device="cuda")
        L.check(L.lib().ds_colsum(L.ptr(dyxn), L.ptr(dg), 1, M, D, D, 0, 0, L.stream()))
        L.check(L.lib().ds_colsum(L.ptr(dyc), L.ptr(db_), 1, M, D, D, 0, 1, L.stream()))     # accumulate into zeros
        assert close(dg.cpu()[0], gamma.grad) and close(db_.cpu()[0], beta.grad)


@pytest.mark.parametrize("G,R,C", [(1, 5300, 1024), (20, 265, 1024), (1, 795, 4096), (3, 70, 257), (1, 5, 300)])
def test_colsum_chunked(L, G, R, C):
    """ds_colsum_ws (row chunks summed by separate workgroups, then added in a fixed order) against float64 and against a
    second run (deterministic), plain and accumulating."""
    x = rnd((G * R, C), "csw.%d.%d" % (R, C), 2.0)
    ref = x.double().view(G, R, C).sum(1)
    xc = x.cuda()
    work = torch.empty(G * 64 * C, device="cuda")
    outs = []
    for _ in range(2):
        out = torch.full((G, C), float("nan"), device="cuda")
        L.check(L.lib().ds_colsum_ws(L.ptr(xc), L.ptr(out), G, R, C, C, R * C, 0, L.ptr(work), work.numel(), L.stream()))
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    acc = torch.ones(G, C, device="cuda")
    L.check(L.lib().ds_colsum_ws(L.ptr(xc), L.ptr(acc), G, R, C, C, R * C, 1, L.ptr(work), work.numel(), L.stream()))
    assert (acc.cpu().double() - 1.0 - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("G,R,C", [(1, 83, 1024), (1, 83, 4096), (2, 18, 100), (1, 16, 64), (3, 37, 1000)])
def test_colsum_small_launch_keeps_the_chain_order(L, G, R, C):
    """ds_colsum on few, long columns (the bias gradients: 83 row-tile partials) runs the four interleaved chains of a column on four
    threads: the result is bit-for-bit the one-thread order  (s0 + s1) + (s2 + s3),  s_j = rows j, j + 4, ... in turn, the R % 4
    tail rows joining s0 -- reproduced here in fp32 on the host."""
    x = rnd((G * R, C), "cs4.%d.%d" % (R, C), 2.0)
    xs = x.view(G, R, C)
    R4 = R - R % 4
    ch = [torch.zeros(G, C) for _ in range(4)]
    for r in range(R4):
        ch[r % 4] = ch[r % 4] + xs[:, r]
    for r in range(R4, R):
        ch[0] = ch[0] + xs[:, r]
    want = (ch[0] + ch[1]) + (ch[2] + ch[3])
    xc = x.cuda()
    out = torch.full((G, C), float("nan"), device="cuda")
    L.check(L.lib().ds_colsum(L.ptr(xc), L.ptr(out), G, R, C, C, R * C, 0, L.stream()))
    assert torch.equal(out.cpu(), want)
    acc = torch.ones(G, C, device="cuda")
    L.check(L.lib().ds_colsum(L.ptr(xc), L.ptr(acc), G, R, C, C, R * C, 1, L.stream()))
    assert torch.equal(acc.cpu(), 1.0 + want)


@pytest.mark.parametrize("G,B,K,D", [(3, 20, 2048, 1024), (1, 32, 256, 300), (2, 1, 512, 64)])
def test_rows_times_row_major_matrix(L, G, B, K, D):
    """ds_rows_times_matrix: B <= 32 rows times G row-major [K][D] matrices (the AdaLN backward's (d modulation) x linear.weight),
    K / 256 partial results added by ds_colsum -- against float64, twice (deterministic), columns past D untouched."""
    x, W = rnd((G, B, K), "rtm.x%d" % K, 2.0), rnd((G, K, D), "rtm.w%d" % D, 0.5)
    ref = torch.einsum("gbk,gkd->gbd", x.double(), W.double())
    xc, Wc = x.cuda(), W.cuda()
    KS = K // 256
    outs = []
    for _ in range(2):
        part = torch.full((KS, G, B, D), float("nan"), device="cuda")
        L.check(L.lib().ds_rows_times_matrix(L.ptr(xc), L.ptr(Wc), L.ptr(part), G, B, K, D, L.stream()))
        out = torch.empty(G, B, D, device="cuda")
        L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(out), 1, KS, G * B * D, G * B * D, 0, 0, L.stream()))
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert L.lib().ds_rows_times_matrix(L.ptr(xc), L.ptr(Wc), L.ptr(part), G, 33, K, D, L.stream()) != 0      # more than 32 rows
    assert L.lib().ds_rows_times_matrix(L.ptr(xc), L.ptr(Wc), L.ptr(part), G, B, K + 32, D, L.stream()) != 0  # K % 256


@pytest.mark.parametrize("G,B,N,D", [(3, 20, 2048, 1024), (1, 32, 50, 2052), (2, 1, 33, 64)])
def test_sum_of_outer_products(L, G, B, N, D):
    """ds_rows_outer: out[g] = a[g]^T s[g] over B <= 32 rows (the AdaLN backward's d linear.weight) against float64; rows of the
    output past N are not written."""
    a, s_ = rnd((G, B, N), "ro.a%d" % N, 2.0), rnd((G, B, D), "ro.s%d" % D, 0.5)
    ref = torch.einsum("gbn,gbd->gnd", a.double(), s_.double())
    ac, sc = a.cuda(), s_.cuda()
    out = torch.full((G * N + 3, D), float("nan"), device="cuda")
    L.check(L.lib().ds_rows_outer(L.ptr(ac), L.ptr(sc), L.ptr(out), G, B, N, D, L.stream()))
    got = out[:G * N].view(G, N, D).cpu().double()
    assert (got - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert torch.isnan(out[G * N:]).all()
    assert L.lib().ds_rows_outer(L.ptr(ac), L.ptr(sc), L.ptr(out), G, 33, N, D, L.stream()) != 0


def test_gelu2_forward_backward(L):
    x = rnd((300, 4096), "g2.x", 6.0).double().requires_grad_(True)
    dy = rnd((300, 4096), "g2.dy")
    y = x * torch.sigmoid(1.702 * x)
    y.backward(dy.double())
    xc, dyc = x.detach().float().cuda(), dy.cuda()
    out = torch.empty_like(xc)
    L.check(L.lib().ds_gelu2(L.ptr(xc), None, L.ptr(out), xc.numel(), L.stream()))
    assert close(out.cpu(), y.detach())
    L.check(L.lib().ds_gelu2(L.ptr(xc), L.ptr(dyc), L.ptr(out), xc.numel(), L.stream()))
    assert close(out.cpu(), x.grad)


def test_softmax_backward_rows(L):
    rows, n, ld = 2 * 16 * 265, 265, 288
    s = rnd((rows, n), "sb.s", 4.0).double().requires_grad_(True)
    dP = rnd((rows, n), "sb.dp")
    P = torch.softmax(s * 0.125, dim=1)
    P.backward(dP.double())
    Pc = torch.zeros(rows, ld, device="cuda")
    Pc[:, :n] = P.detach().float().cuda()
    dPc = torch.full((rows, ld), 7.0, device="cuda")
    dPc[:, :n] = dP.cuda()
    L.check(L.lib().ds_softmax_bwd_rows(L.ptr(Pc), L.ptr(dPc), rows, n, ld, 0.125, L.stream()))
    assert close(dPc.cpu()[:, :n], s.grad) and (dPc[:, n:] == 0).all()


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("B,H,Lq,Lk,fusedkv", [(2, 16, 265, 265, 3), (2, 16, 265, 77, 2), (1, 2, 40, 33, 2), (3, 4, 288, 96, 2)])
def test_attention_backward_recompute(L, B, H, Lq, Lk, fusedkv, split):
    """ds_attention_bwd / ds_attention_bwd_f16x2 (two kernels, tile-wise recomputation, nothing stored between forward and
    backward but O): dQ / dK / dV written in place into fused projection gradients, against float64 autograd of
    softmax(q k^T / 8) v; the forward it differentiates is ds_attention on the same in-place operands.  split: every tile
    product on the fp16 matrix cores (3-pass split) with dO at the magnitude the training step's loss scale gives it."""
    D = H * 64
    if fusedkv == 3:            # self-attention: q | k | v columns of one [B*L][3D] buffer
        qkv = rnd((B * Lq, 3 * D), "ab.qkv.%d" % Lq, 1.5)
        srcs = [(qkv, 0, 3 * D), (qkv, D, 3 * D), (qkv, 2 * D, 3 * D)]
    else:                       # cross-attention: q [B*Lq][D], k | v columns of [B*Lk][2D]
        q, kv = rnd((B * Lq, D), "ab.q.%d" % Lq, 1.5), rnd((B * Lk, 2 * D), "ab.kv.%d" % Lk, 1.5)
        srcs = [(q, 0, D), (kv, 0, 2 * D), (kv, D, 2 * D)]
    dO = rnd((B * Lq, D), "ab.do.%d" % Lq, 0.7 * (4096.0 if split else 1.0))      # (split: |dO| up to 2^11.5, as under the loss scale)

    def heads(t, col, Lx):
        return t[:, col:col + D].reshape(B, Lx, H, 64).permute(0, 2, 1, 3)
    Q, K, V = (heads(t.double(), col, Lx).clone().requires_grad_(True) for (t, col, _), Lx in zip(srcs, (Lq, Lk, Lk)))
    out = torch.softmax(0.125 * (Q @ K.transpose(-1, -2)), dim=-1) @ V
    out.backward(heads(dO.double(), 0, Lq))
    dev = {id(t): t.cuda() for t, _, _ in srcs}
    ops = [(dev[id(t)], col, ld) for t, col, ld in srcs]
    o = torch.empty(B * Lq, D, device="cuda")
    L.check(L.lib().ds_attention(L.ptr_off(*ops[0][:2]), ops[0][2], L.ptr_off(*ops[1][:2]), ops[1][2], L.ptr_off(*ops[2][:2]),
                                 ops[2][2], L.ptr(o), D, B, H, Lq, Lk, 0.125, L.stream()))
    want_o = out.detach().permute(0, 2, 1, 3).reshape(B * Lq, D)
    assert close(o.cpu(), want_o, 1e-5)
    grads = {id(t): torch.full(t.shape, float("nan"), device="cuda") for t, _, _ in srcs}
    gops = [(grads[id(t)], col, ld) for t, col, ld in srcs]
    stats = torch.empty(2 * B * H * ((Lq + 31) // 32 * 32), device="cuda")
    dOc = dO.cuda()
    bwd = L.lib().ds_attention_bwd_f16x2 if split else L.lib().ds_attention_bwd
    L.check(bwd(
        L.ptr_off(*ops[0][:2]), ops[0][2], L.ptr_off(*ops[1][:2]), ops[1][2], L.ptr_off(*ops[2][:2]), ops[2][2], L.ptr(o), D,
        L.ptr(dOc), D, L.ptr_off(*gops[0][:2]), gops[0][2], L.ptr_off(*gops[1][:2]), gops[1][2], L.ptr_off(*gops[2][:2]), gops[2][2],
        L.ptr(stats), B, H, Lq, Lk, 0.125, L.stream()))
    for name, (gt, col, _), ref, Lx in zip("qkv", gops, (Q, K, V), (Lq, Lk, Lk)):
        got = gt.cpu()[:, col:col + D]
        want = ref.grad.permute(0, 2, 1, 3).reshape(B * Lx, D)
        err = (got.double() - want).abs().max().item() / want.abs().max().item()
        print("d%s: rel err %.2e" % (name, err))
        assert err < 2e-5, (name, err)
    for gt in grads.values():
        assert not torch.isnan(gt).any()        # every column range of the fused gradient buffers was written
    if split:
        # ds_attention_bwd_f16x2_mon (round 6): the same kernels fold max |dO| -- the operand that carries the step's loss
        # scale into their fp16 splits -- into the saturation monitor.  Same gradients bit for bit, and the scalar equals the
        # float64 maximum (through the fp16 hi plane: 2^-11 relative).
        want_m = float(dO.abs().max())
        grads2 = {id(t): torch.full(t.shape, float("nan"), device="cuda") for t, _, _ in srcs}
        gops2 = [(grads2[id(t)], col, ld) for t, col, ld in srcs]
        for start in (0.0, 2.0 * want_m):       # an empty monitor takes the maximum; one that already holds more is left alone
            amax = torch.full((1,), start, device="cuda")
            L.check(L.lib().ds_attention_bwd_f16x2_mon(
                L.ptr_off(*ops[0][:2]), ops[0][2], L.ptr_off(*ops[1][:2]), ops[1][2], L.ptr_off(*ops[2][:2]), ops[2][2], L.ptr(o), D,
                L.ptr(dOc), D, L.ptr_off(*gops2[0][:2]), gops2[0][2], L.ptr_off(*gops2[1][:2]), gops2[1][2], L.ptr_off(*gops2[2][:2]),
                gops2[2][2], L.ptr(stats), B, H, Lq, Lk, 0.125, 1.0, L.ptr(amax), L.stream()))
            got_m = float(amax.item())
            assert abs(got_m - max(start, want_m)) <= 1e-3 * max(start, want_m), (got_m, want_m)
        for key in grads:
            assert torch.equal(grads[key], grads2[key])


@pytest.mark.parametrize("case", ["dS above fp16's range", "dS 2^-20 under dO", "dO at 2^-13 with the call's own scale"])
def test_attention_backward_dS_is_normalised_per_wave(L, case):
    """dS = scale P (dP - delta) exists only in registers and is split to fp16 for the dQ / dK products.  It has no fixed relation
    to dO: (a) with |V| ~ 40 and a large dO it exceeds 65504 -- the split would SATURATE silently; (b) with
    near-uniform probabilities and nearly equal value rows it is the small difference of two nearly equal numbers, ~2^-20 of
    dO -- its fp16 planes would sit in the subnormal range (the 19-layer B = 20 golden found 1e-2 errors in the cross-attention
    query gradients that way).  Round 6: the wave's dS tiles are normalised by an exact power of two before the split and the
    dQ / dK store takes it out again: both cases come out fp32-class against float64.  (c): dO itself 2^-13 -- what a deep
    layer's attention sees when the loss scale is set elsewhere -- with the call's own power of two `do_scale` (the training step
    calibrates one per attention backward): dO * do_scale is what is split, the stores take it out again."""
    B, H, Lq, Lk = 1, 2, 72, 77
    D = H * 64
    if case.startswith("dS above"):
        q, kv = rnd((B * Lq, D), "abn.q", 1.5), rnd((B * Lk, 2 * D), "abn.kv", 1.5)
        kv[:, D:] *= 40.0
        dO = rnd((B * Lq, D), "abn.do", 10000.0)
    elif case.startswith("dO at"):         # (c) a dO far under the loss scale's place: the call's own do_scale brings it back
        q, kv = rnd((B * Lq, D), "abn.q", 1.5), rnd((B * Lk, 2 * D), "abn.kv", 1.5)
        dO = rnd((B * Lq, D), "abn.do", 2.0 ** -15)
    else:
        q, kv = rnd((B * Lq, D), "abn.q", 0.02), rnd((B * Lk, 2 * D), "abn.kv", 0.02)
        base = rnd((1, D), "abn.v0", 1.0)
        kv[:, D:] = base + 1e-4 * rnd((B * Lk, D), "abn.vn", 1.0)          # value rows equal to 1e-4
        dO = rnd((B * Lq, D), "abn.do", 64.0)

    def heads(t, col, Lx):
        return t[:, col:col + D].reshape(B, Lx, H, 64).permute(0, 2, 1, 3).double()
    Q, K, V = (heads(t, c, Lx).clone().requires_grad_(True) for t, c, Lx in ((q, 0, Lq), (kv, 0, Lk), (kv, D, Lk)))
    dOh = heads(dO, 0, Lq)
    P = torch.softmax(0.125 * (Q @ K.transpose(-1, -2)), dim=-1)
    out = P @ V
    out.backward(dOh)
    dS = 0.125 * P.detach() * (dOh @ V.detach().transpose(-1, -2) - (dOh * out.detach()).sum(-1, keepdim=True))
    ratio = float(dS.abs().max()) / float(dO.abs().max())
    do_scale = 1.0
    if case.startswith("dS above"):
        assert float(dS.abs().max()) > 65504.0
    elif case.startswith("dO at"):
        do_scale = 2.0 ** 19
        assert float(dO.abs().max()) < 2.0 ** -12
    else:
        assert ratio < 2.0 ** -18
    qc, kvc, dOc = q.cuda(), kv.cuda(), dO.cuda()
    o = torch.empty(B * Lq, D, device="cuda")
    L.check(L.lib().ds_attention(L.ptr(qc), D, L.ptr(kvc), 2 * D, L.ptr_off(kvc, D), 2 * D, L.ptr(o), D, B, H, Lq, Lk, 0.125, L.stream()))
    dq, dkv = torch.empty_like(qc), torch.empty_like(kvc)
    stats = torch.empty(2 * B * H * 96, device="cuda")
    amax = torch.zeros(1, device="cuda")
    L.check(L.lib().ds_attention_bwd_f16x2_mon(L.ptr(qc), D, L.ptr(kvc), 2 * D, L.ptr_off(kvc, D), 2 * D, L.ptr(o), D, L.ptr(dOc), D,
                                               L.ptr(dq), D, L.ptr(dkv), 2 * D, L.ptr_off(dkv, D), 2 * D, L.ptr(stats), B, H, Lq, Lk,
                                               0.125, do_scale, L.ptr(amax), L.stream()))
    unheads = lambda g_, Lx: g_.permute(0, 2, 1, 3).reshape(B * Lx, D)
    errs = {}
    for name, got, want in (("dq", dq.cpu(), unheads(Q.grad, Lq)), ("dk", dkv.cpu()[:, :D], unheads(K.grad, Lk)),
                            ("dv", dkv.cpu()[:, D:], unheads(V.grad, Lk))):
        errs[name] = float((got.double() - want).abs().max() / want.abs().max())
    # the yardstick: the same formulas evaluated in fp32 by torch on the CPU
    Qf, Kf, Vf = (x.detach().float().requires_grad_(True) for x in (Q, K, V))
    (torch.softmax(0.125 * (Qf @ Kf.transpose(-1, -2)), dim=-1) @ Vf).backward(dOh.float())
    ref = {n: float((g_.grad.double() - w.grad).abs().max() / w.grad.abs().max()) for n, g_, w in (("dq", Qf, Q), ("dk", Kf, K), ("dv", Vf, V))}
    print("%s: max |dS| / max |dO| = 2^%.1f; rel err vs float64 %s (torch fp32 on the same formulas: %s)"
          % (case, math.log2(ratio), {k: "%.1e" % v for k, v in errs.items()}, {k: "%.1e" % v for k, v in ref.items()}))
    assert abs(float(amax.item()) - do_scale * float(dO.abs().max())) < 1e-3 * do_scale * float(dO.abs().max())
    for n in errs:
        assert errs[n] < max(2e-5, 20 * ref[n]), (n, errs[n], ref[n])


def test_embedding_backward_and_colsum_strided(L):
    M, D, rows = 530, 1024, 257
    tok = synth.synth_tokens(2, 265, 256, mask_frac=0.3, key="eb.t").view(-1)
    dx = rnd((M, D), "eb.dx")
    ref = torch.zeros(rows, D, dtype=torch.float64).index_add_(0, tok, dx.double())
    demb = torch.zeros(rows, D, device="cuda")
    dxc, tokc = dx.cuda(), tok.cuda()
    L.check(L.lib().ds_embed_bwd(L.ptr(dxc), L.ptr(tokc), L.ptr(demb), M, D, rows, L.stream()))
    assert close(demb.cpu(), ref, 1e-5)
    # position table gradient: sum over the batch for every grid position = column sums with a sample stride
    dpos = torch.empty(265, D, device="cuda")
    L.check(L.lib().ds_colsum(L.ptr(dxc), L.ptr(dpos), 265, 2, D, 265 * D, D, 0, L.stream()))
    assert close(dpos.cpu(), dx.view(2, 265, D).double().sum(0))


def test_adamw_matches_torch(L):
    n = 100003
    p0, g1, g2 = rnd((n,), "aw.p"), rnd((n,), "aw.g1", 0.1), rnd((n,), "aw.g2", 0.1)
    ref = p0.clone().double().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step, g in enumerate((g1, g2), start=1):
        ref.grad = g.double()
        opt.step()
        gc = g.cuda()
        L.check(L.lib().ds_adamw(L.ptr(p), L.ptr(gc), L.ptr(m), L.ptr(v), n, 3e-3, 0.9, 0.96, 1e-8, 4.5e-2, step, L.stream()))
    assert close(p.cpu(), ref.detach(), 2e-6)


def test_gemm_f16x2_split_k_groups(L):
    """The row-major form of a split-K launch (fp32 A split by the loader, row-major fp16-plane W -- the training step itself
    uses the packed form below): C = A W^T with the contraction split into `groups` K-ranges of one grouped ds_gemm_f16x2
    launch, partial products summed by ds_colsum -- against float64 and against the ungrouped launch."""
    N, K, Mp, S = 320, 192, 1024, 4
    a = rnd((N, Mp), "sk.a", 30.0)
    x = rnd((K, Mp), "sk.x", 2.0)
    ref = a.double() @ x.double().t()
    ac, xc = a.cuda(), x.cuda()
    planes = _split_planes(x).view(torch.int16).cuda()          # row-major hi | lo fp16 planes [2][K][Mp]
    one = torch.empty(N, K, device="cuda")
    L.gemm(ac, planes, one, N, K, Mp, split2=0.5, w_plane=K * Mp)
    part = torch.full((S, N * K), float("nan"), device="cuda")
    Kc = Mp // S
    L.gemm(ac, planes, part, N, K, Kc, lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc, w_gstride=Kc, c_gstride=N * K,
           split2=0.5, w_plane=K * Mp)
    out = torch.empty(N, K, device="cuda")
    L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(out), 1, S, N * K, N * K, 0, 0, L.stream()))
    scale = ref.abs().max().item()
    assert (one.cpu().double() - 0.5 * ref).abs().max().item() < 2e-6 * scale
    assert (out.cpu().double() - 0.5 * ref).abs().max().item() < 2e-6 * scale
    for g in range(S):       # every group is the product over its own K-range
        pr = a[:, g * Kc:(g + 1) * Kc].double() @ x[:, g * Kc:(g + 1) * Kc].double().t()
        assert (part[g].view(N, K).cpu().double() - 0.5 * pr).abs().max().item() < 2e-6 * scale


def _split_planes(a):
    """ds_split_hi / ds_split_lo (csrc/common.h) in torch: [2][R][K] fp16"""
    hi = a.clamp(-65504.0, 65504.0).half()
    lo = (a - hi.float()).clamp(-65504.0, 65504.0).half()
    return torch.stack((hi, lo)).contiguous()


@pytest.mark.parametrize("rows,cols,S", [(531, 1024, 4), (1540, 512, 8), (70, 96, 1), (5300, 256, 8)])
def test_pack_operand_all_outputs(L, rows, cols, S):
    """ds_pack_operand (csrc/pack.hip): one pass -> packed row form, packed transposed form with the contraction padded
    for a split-K launch, per-tile column sums and max |x| -- each bit for bit what torch computes for the same definition
    (the host packer _lib.pack_planes is the layout the GEMM tests already pin)."""
    ld = cols + 8
    src = torch.full((rows, ld), 123.0)
    src[:, :cols] = rnd((rows, cols), "pk.src%d" % rows, 40.0)
    src[0, 0], src[1, 1], src[2, 2] = 7.0e4, -9.9e9, 3.0e-7            # saturation and the subnormal low plane
    scale = 2.0 ** 3
    want = src[:, :cols] * scale
    rows_pad = (rows + 32 * S - 1) // (32 * S) * (32 * S)
    dev = src.cuda()
    R16, C16 = (rows + 15) // 16 * 16, (cols + 15) // 16 * 16
    drow = torch.full((2, R16 * cols), 0x7777, dtype=torch.int16, device="cuda")
    dt = torch.full((2, C16 * rows_pad), 0x7777, dtype=torch.int16, device="cuda")
    tr = L.lib().ds_pack_operand_tile_rows(rows, rows_pad)
    part = torch.full((tr, cols), float("nan"), device="cuda")
    amax = torch.zeros(1, device="cuda")
    L.check(L.lib().ds_pack_operand(L.ptr(dev), rows, cols, ld, scale, 0, None, 0, L.ptr(drow), R16 * cols, L.ptr(dt),
                                    C16 * rows_pad, rows_pad, 0, 0, L.ptr(part), L.ptr(amax), L.stream()))
    got_row = L.unpack_planes(drow.cpu().view(torch.float16).view(2, -1), R16, cols)
    full = torch.zeros(R16, cols)
    full[:rows] = want
    assert torch.equal(got_row, _split_planes(full)), "row form"
    got_t = L.unpack_planes(dt.cpu().view(torch.float16).view(2, -1), C16, rows_pad)
    fullt = torch.zeros(C16, rows_pad)
    fullt[:cols, :rows] = want.t()
    assert torch.equal(got_t, _split_planes(fullt)), "transposed form (zero-padded contraction)"
    nr = (rows + 63) // 64
    sums = part[:nr].cpu().double().sum(0)            # (round 6: the column sums are those of X itself, BEFORE `scale`)
    unscaled = src[:, :cols].double()
    assert (sums - unscaled.sum(0)).abs().max().item() < 1e-5 * unscaled.abs().sum(0).max().item()
    want_amax = want.abs().max().item()
    assert amax.item() == want_amax


def test_pack_operand_parts_into_a_fused_operand(L):
    """The sub-range form: query | key | value weights packed straight into their row groups (row form) and k-ranges
    (transposed form) of ONE fused operand == packing the concatenated matrix (what the training step did with torch.cat)."""
    K, Ns = 256, (64, 96, 32)
    parts = [rnd((n, K), "pkp.w%d" % i, 0.3).cuda() for i, n in enumerate(Ns)]
    N = sum(Ns)
    fused = torch.cat(parts)
    K16 = (K + 15) // 16 * 16

    def alloc():
        return (torch.full((2, N * K), 0x7777, dtype=torch.int16, device="cuda"),
                torch.full((2, K16 * N), 0x7777, dtype=torch.int16, device="cuda"))
    row_a, t_a = alloc()
    L.check(L.lib().ds_pack_operand(L.ptr(fused), N, K, K, 4.0, 0, None, 0, L.ptr(row_a), N * K, L.ptr(t_a), K16 * N, N, 0, 0, None, None,
                                    L.stream()))
    row_b, t_b = alloc()
    n0 = 0
    for w in parts:
        n = w.shape[0]
        L.check(L.lib().ds_pack_operand(L.ptr(w), n, K, K, 4.0, 0, None, 0, L.ptr_off(row_b, n0 * K), N * K, L.ptr(t_b), K16 * N, n, n0, N,
                                        None, None, L.stream()))
        n0 += n
    assert torch.equal(row_a, row_b) and torch.equal(t_a, t_b)
    assert int((row_a == 0x7777).sum()) < 8 and int((t_a == 0x7777).sum()) < 8        # (everything was written)


@pytest.mark.parametrize("cfg", [0, 1, 3])
@pytest.mark.parametrize("S", [1, 2])
def test_gemm_f16x2_multi_equals_separate_launches(L, cfg, S):
    """ds_gemm_f16x2_multi: up to four independent dW = dY^T X products (different shapes, the same number of K-ranges) as ONE
    grid of a tile configuration == each product launched alone with that tile forced, bit for bit; five products, unequal
    K-range counts and a residual are refused."""
    M = 530
    shapes = [(256, 128), (128, 384), (96, 96), (512, 64)]
    Mp = (M + 32 * S - 1) // (32 * S) * (32 * S)
    Kc = Mp // S

    def tform(src, cols):
        C16 = (cols + 15) // 16 * 16
        d = torch.empty(2, C16 * Mp, dtype=torch.int16, device="cuda")
        L.check(L.lib().ds_pack_operand(L.ptr(src), M, cols, cols, 1.0, 0, None, 0, None, 0, L.ptr(d), C16 * Mp, Mp, 0, 0, None, None,
                                        L.stream()))
        return d, C16 * Mp
    keep, descs, outs, refs = [], [], [], []
    for i, (N, K) in enumerate(shapes):
        dy, x = rnd((M, N), "gm.dy%d" % i, 2.0).cuda(), rnd((M, K), "gm.x%d" % i, 1.5).cuda()
        a, apl = tform(dy, N)
        w, wpl = tform(x, K)
        keep += [a, w]
        kw = dict(lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16, c_gstride=N * K, split2=0.25, a_plane=apl, w_plane=wpl)
        ref = torch.full((S, N * K), float("nan"), device="cuda")
        L.lib().ds_gemm_f16x2_force_tile(cfg)
        try:
            L.gemm(a, w, ref, N, K, Kc, **kw)
        finally:
            L.lib().ds_gemm_f16x2_force_tile(-1)
        out = torch.full((S, N * K), float("nan"), device="cuda")
        descs.append(L.gemm(a, w, out, N, K, Kc, desc_only=True, **kw))
        outs.append(out)
        refs.append(ref)
    L.gemm_multi(descs, cfg)
    for out, ref in zip(outs, refs):
        assert torch.equal(out, ref)
    L.gemm_multi(descs[:1], cfg)                      # one product is fine too
    with pytest.raises(RuntimeError):
        L.gemm_multi(descs + descs[:1], cfg)          # five
    with pytest.raises(RuntimeError):
        L.gemm_multi(descs, 2)                        # no 64 x 64 form
    bad = L.gemm(keep[0], keep[1], outs[0], shapes[0][0], shapes[0][1], Kc, desc_only=True, lda=Mp, ldw=Mp, ldc=shapes[0][1], groups=S + 1,
                 a_gstride=Kc * 16, w_gstride=Kc * 16, c_gstride=shapes[0][0] * shapes[0][1], split2=0.25, a_plane=descs[0].a_plane,
                 w_plane=descs[0].w3_plane)
    with pytest.raises(RuntimeError):
        L.gemm_multi([descs[1], bad], cfg)            # unequal K-range counts


def test_pack_operand_gelu_prologues(L):
    """The MLP's activation rides in the pack: DS_PACK_GELU2 (x := gelu2(x), transformer_utils.py:111-115) and
    DS_PACK_GELU2_BWD (x := x * gelu2'(aux)) against float64; the packed value (hi + lo) is the fp32 result to 2^-22."""
    rows, cols = 300, 128
    u = rnd((rows, cols), "pkg.u", 4.0)
    dy = rnd((rows, cols), "pkg.dy", 2.0)
    uc, dyc = u.cuda(), dy.cuda()
    R16 = (rows + 15) // 16 * 16
    for pro, src, aux, ref in ((1, uc, None, u.double() * torch.sigmoid(1.702 * u.double())),
                               (2, dyc, uc, dy.double() * (torch.sigmoid(1.702 * u.double()) *
                                                           (1 + 1.702 * u.double() * (1 - torch.sigmoid(1.702 * u.double())))))):
        d = torch.empty(2, R16 * cols, dtype=torch.int16, device="cuda")
        L.check(L.lib().ds_pack_operand(L.ptr(src), rows, cols, cols, 1.0, pro, L.ptr(aux), cols, L.ptr(d), R16 * cols, None, 0, 0, 0, 0,
                                        None, None, L.stream()))
        pl = L.unpack_planes(d.cpu().view(torch.float16).view(2, -1), rows, cols)
        val = pl[0].double() + pl[1].double()
        assert (val - ref).abs().max().item() < 3e-6 * ref.abs().max().item()


def test_gemm_f16x2_packed_split_k_groups(L):
    """The dW launch of round 5: BOTH operands as packed planes (ds_pack_operand's transposed forms), the contraction split
    into `groups` K-ranges of one grouped launch (row groups lda / 32 k-tiles apart, group stride 16 K halves), partial
    products summed by ds_colsum -- against float64, against the ungrouped packed launch and per K-range."""
    N, K, M, S = 320, 192, 1000, 4
    Mp = (M + 32 * S - 1) // (32 * S) * (32 * S)
    dy = rnd((M, N), "psk.dy", 30.0)
    x = rnd((M, K), "psk.x", 2.0)
    ref = dy.double().t() @ x.double()
    dyc, xc = dy.cuda(), x.cuda()

    def tform(src, cols):
        C16 = (cols + 15) // 16 * 16
        d = torch.empty(2, C16 * Mp, dtype=torch.int16, device="cuda")
        L.check(L.lib().ds_pack_operand(L.ptr(src), M, cols, cols, 1.0, 0, None, 0, None, 0, L.ptr(d), C16 * Mp, Mp, 0, 0, None, None,
                                        L.stream()))
        return d, C16 * Mp
    a, apl = tform(dyc, N)
    w, wpl = tform(xc, K)
    one = torch.empty(N, K, device="cuda")
    L.gemm(a, w, one, N, K, Mp, split2=0.5, a_plane=apl, w_plane=wpl)
    scale = ref.abs().max().item()
    assert (one.cpu().double() - 0.5 * ref).abs().max().item() < 2e-6 * scale
    Kc = Mp // S
    for tile in (-1, 0, 1, 2, 3):
        L.lib().ds_gemm_f16x2_force_tile(tile)
        try:
            part = torch.full((S, N * K), float("nan"), device="cuda")
            L.gemm(a, w, part, N, K, Kc, lda=Mp, ldw=Mp, ldc=K, groups=S, a_gstride=Kc * 16, w_gstride=Kc * 16, c_gstride=N * K,
                   split2=0.5, a_plane=apl, w_plane=wpl)
        finally:
            L.lib().ds_gemm_f16x2_force_tile(-1)
        out = torch.empty(N, K, device="cuda")
        L.check(L.lib().ds_colsum(L.ptr(part), L.ptr(out), 1, S, N * K, N * K, 0, 0, L.stream()))
        assert (out.cpu().double() - 0.5 * ref).abs().max().item() < 2e-6 * scale, "tile %d" % tile
        for g in range(S):       # every group is the product over its own K-range (the last one includes the zero padding)
            lo_, hi_ = g * Kc, min((g + 1) * Kc, M)
            pr = dy[lo_:hi_].double().t() @ x[lo_:hi_].double() if hi_ > lo_ else torch.zeros(N, K, dtype=torch.float64)
            assert (part[g].view(N, K).cpu().double() - 0.5 * pr).abs().max().item() < 2e-6 * scale, "tile %d group %d" % (tile, g)


def test_adamw_multi_equals_per_tensor_launches(L):
    """ds_adamw_multi (64 tensor descriptors by value per launch) against ds_adamw_dev tensor by tensor (the same expressions;
    hipcc contracts the two kernels' multiply-adds differently, so equal to rounding, not bit for bit): odd sizes, an
    unaligned view, more tensors than one batch; elements next to a tensor are not touched."""
    import ctypes
    import math
    sizes = [50001, 7, 4096, 1, 123457] + [33 + 17 * i for i in range(70)]
    hyper = torch.tensor([3e-3, 1.0 - 0.9 ** 2, math.sqrt(1.0 - 0.96 ** 2), 0.25], device="cuda")
    base = [rnd((n + 1,), "awm.p%d" % i).cuda() for i, n in enumerate(sizes)]
    ps = [b[1:] if i == 1 else b[:-1] for i, b in enumerate(base)]                 # tensor 1: a 4-byte-offset view
    gs = [rnd((n,), "awm.g%d" % i, 0.1).cuda() for i, n in enumerate(sizes)]
    ms = [rnd((n,), "awm.m%d" % i, 0.01).cuda() for i, n in enumerate(sizes)]
    vs = [rnd((n,), "awm.v%d" % i, 0.01).abs().cuda() for i, n in enumerate(sizes)]
    pr, mr, vr = [p.clone() for p in ps], [m.clone() for m in ms], [v.clone() for v in vs]
    for p, g, m, v in zip(pr, gs, mr, vr):
        L.check(L.lib().ds_adamw_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(hyper), 0.9, 0.96, 1e-8, 4.5e-2, L.stream()))
    rec = (ctypes.c_int64 * (5 * len(sizes)))()
    for i, (p, g, m, v) in enumerate(zip(ps, gs, ms, vs)):
        rec[5 * i:5 * i + 5] = [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()]
    L.check(L.lib().ds_adamw_multi(ctypes.cast(rec, ctypes.c_void_p), len(sizes), L.ptr(hyper), 0.9, 0.96, 1e-8, 4.5e-2, L.stream()))
    torch.cuda.synchronize()
    for i in range(len(sizes)):
        assert close(ps[i].cpu(), pr[i].cpu(), 1e-6) and close(ms[i].cpu(), mr[i].cpu(), 1e-6) and close(vs[i].cpu(), vr[i].cpu(), 1e-6), \
            "tensor %d" % i
        assert float(base[i][0 if i == 1 else -1]) == float(rnd((sizes[i] + 1,), "awm.p%d" % i)[0 if i == 1 else -1])   # neighbours untouched


def test_amax_and_adamw_dev(L):
    x = rnd((100003,), "am.x", 3.0)
    x[4711] = -17.25
    out = torch.zeros(1, device="cuda")
    xc = x.cuda()
    L.check(L.lib().ds_amax(L.ptr(xc), xc.numel(), L.ptr(out), L.stream()))
    assert out.item() == 17.25
    n = 50001
    p0, g1 = rnd((n,), "awd.p"), rnd((n,), "awd.g", 0.1)
    pa, ma, va = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb, mb, vb = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    gc = g1.cuda()
    g_scaled = (gc * 0.25).contiguous()
    import math
    for step in (1, 2):
        L.check(L.lib().ds_adamw(L.ptr(pa), L.ptr(g_scaled), L.ptr(ma), L.ptr(va), n, 3e-3, 0.9, 0.96, 1e-8, 4.5e-2, step, L.stream()))
        hyper = torch.tensor([3e-3, 1.0 - 0.9 ** step, math.sqrt(1.0 - 0.96 ** step), 0.25], device="cuda")
        L.check(L.lib().ds_adamw_dev(L.ptr(pb), L.ptr(gc), L.ptr(mb), L.ptr(vb), n, L.ptr(hyper), 0.9, 0.96, 1e-8, 4.5e-2, L.stream()))
    assert close(pb.cpu(), pa.cpu(), 1e-6) and close(mb.cpu(), ma.cpu(), 1e-6)


@pytest.mark.parametrize("precision,attention", [("fp32", "fused"), ("f16x2", "fused"), ("f16x2", "composed")])
def test_training_step_gradients_vs_oracle_autograd(precision, attention):
    """The whole backward of the denoiser on the HIP kernels (modeling/train.py; linear layers on the exact-fp32 MFMA or
    on the 3-pass fp16 split GEMM incl. dX and dW, gradients rescaled by an exact power of two into fp16's range): loss and
    EVERY parameter gradient of the 2-layer model against autograd through the oracle (which the CPU suite pins to
    the reference's loss.backward()), then one AdamW update against torch.optim.AdamW."""
    import diffsound_oracle as O
    from conftest import golden, synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    g = golden("train_loss_L2")
    m = build_model(default_config(n_layer=2, diffusion_step=100))
    sd_cpu = dict(synth_sd("dalle", 2))
    m.load_state_dict({**sd_cpu, **synth_sd("encoder")}, strict=False)
    m = m.cuda().eval()
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0")
    cond = synth.synth_cond_emb(3, key="tl.c")
    t = torch.tensor([57, 0, 93])
    pt = torch.ones(3) / 100
    u = synth.synth_uniform((3, 257, 265), key="tl.u")
    # oracle: loss + autograd gradients on the CPU
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    with torch.enable_grad():
        _, _, loss_ref, _ = O.train_loss(sd, x0, cond, t, pt, u)
        loss_ref.backward()
    step = TrainStep(dt, precision=precision, attention=attention)
    loss, grads = step.loss_and_grads(x0.cuda(), cond.cuda(), t.cuda(), pt.cuda(), u.cuda())
    print("loss %.6f (oracle %.6f, reference %.6f)" % (loss.item(), loss_ref.item(), float(g["loss"])))
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * float(g["loss"])
    worst, missing = [], []
    for k, v in sd.items():
        if not (k.startswith("transformer.transformer.") and v.is_floating_point() and v.grad is not None):
            continue
        name = k[len("transformer."):]
        if name not in grads:
            if v.grad.abs().max() > 0:
                missing.append(name)
            continue
        got, want = grads[name].cpu().double(), v.grad.double()
        if want.abs().max().item() < 1e-7:
            # the key biases have an analytically ZERO gradient (softmax is invariant to them): both sides hold
            # ~1e-9 rounding noise there, which a relative measure cannot compare
            assert got.abs().max().item() < 1e-6, name
            continue
        err = (got - want).abs().max().item() / want.abs().max().item()
        worst.append((err, name, want.abs().max().item()))
    worst.sort(reverse=True)
    print("precision %s: worst per-tensor gradient error %.2e" % (precision, worst[0][0]))
    for err, name, mag in worst[:12]:
        print("  grad rel err %.2e  |g|max %.2e  %s" % (err, mag, name))
    assert not missing, missing
    assert len(worst) >= 50 and worst[0][0] < 2e-3, worst[:5]
    # one optimizer step on a few tensors vs torch.optim.AdamW
    names = ["transformer.to_logits.1.weight", "transformer.blocks.0.attn1.query.weight", "transformer.blocks.1.ln2.weight"]
    params = dict(dt.named_parameters())
    before = {n: params[n].detach().cpu().double().clone() for n in names}
    step.adamw_step({n: grads[n] for n in names}, {}, 1, lr=1e-3)
    for n in names:
        ref = before[n].clone().requires_grad_(True)
        opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2)
        ref.grad = grads[n].cpu().double()
        opt.step()
        assert close(params[n].detach().cpu(), ref.detach(), 2e-6), n


def test_solver_three_iterations_vs_cpu_autograd():
    """The training iteration in the reference's order (modeling/solver.py: gradients -> global-norm clip -> AdamW ->
    EMA) on the HIP training step, three iterations on one batch: every loss and the pre-clip gradient norm against
    the same loop run on the CPU with autograd through the oracle, torch's clip_grad_norm_ and torch.optim.AdamW.  The
    second and third losses depend on the updated weights, so this covers the update path end to end."""
    import diffsound_oracle as O
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import EMA, GradClipWindow, Solver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    m = build_model(default_config(n_layer=2, diffusion_step=100))
    sd_cpu = dict(synth_sd("dalle", 2))
    m.load_state_dict({**sd_cpu, **synth_sd("encoder")}, strict=False)
    m = m.cuda().eval()
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0")
    cond = synth.synth_cond_emb(3, key="tl.c")
    t = torch.tensor([57, 0, 93])
    pt = torch.ones(3) / 100
    u = synth.synth_uniform((3, 257, 265), key="tl.u")
    # CPU loop
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k.startswith("transformer.transformer.") else v)
          for k, v in sd_cpu.items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-3, betas=(0.9, 0.96), eps=1e-8, weight_decay=4.5e-2)
    want = []
    for _ in range(3):
        opt.zero_grad()
        with torch.enable_grad():
            _, _, loss_ref, _ = O.train_loss(sd, x0, cond, t, pt, u)
            loss_ref.backward()
        norm_ref = torch.nn.utils.clip_grad_norm_([p for p in leaves if p.grad is not None], 0.5)
        opt.step()
        want.append((loss_ref.item(), norm_ref.item()))
    # HIP loop
    key = "transformer.to_logits.1.weight"
    w_start = dict(dt.named_parameters())[key].detach().clone()
    solver = Solver(TrainStep(dt), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5),
                    ema=EMA(dt, decay=0.5, update_interval=1, device="cuda"))
    batch = (x0.cuda(), cond.cuda(), t.cuda(), pt.cuda(), u.cuda())
    ema_want = w_start.clone()
    for it in range(3):
        out = solver.step(*batch)
        loss, norm = float(out["loss"]), float(out["grad_norm"])
        print("iter %d: loss %.5f (cpu %.5f)  |g| %.4f (cpu %.4f)" % (it, loss, want[it][0], norm, want[it][1]))
        assert abs(loss - want[it][0]) < 2e-3 * want[it][0]
        assert abs(norm - want[it][1]) < 5e-3 * want[it][1]
        ema_want = ema_want * 0.5 + dict(dt.named_parameters())[key].detach() * 0.5
    assert abs(want[1][0] - want[0][0]) > 0.02 * want[0][0]          # the update visibly moved the loss
    cnt = torch.zeros(100)
    cnt[t] = 3.0
    assert torch.equal(dt.Lt_count.cpu(), cnt) and (dt.Lt_history.cpu()[t] > 0).all()   # sample_time's statistics are kept
    w_end = dict(dt.named_parameters())[key].detach()
    assert not torch.equal(w_end, w_start)
    assert close(solver.ema.state_dict()[key].cpu(), ema_want.cpu(), 1e-6)


def test_graphed_iteration_matches_eager():
    """TrainStep.capture: gradients -> clip -> AdamW of the split-GEMM training step replayed as one hipGraph, three
    iterations with a changing batch (t, noise) and learning rate, against the eager Solver on an identical model."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, GraphSolver, Solver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep

    def make():
        m = build_model(default_config(n_layer=2, diffusion_step=100))
        m.load_state_dict({**dict(synth_sd("dalle", 2)), **synth_sd("encoder")}, strict=False)
        m = m.cuda().eval()
        dt = m.transformer
        dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
        return dt
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0").cuda()
    cond = synth.synth_cond_emb(3, key="tl.c").cuda()
    pt = (torch.ones(3) / 100).cuda()
    batches = [(torch.tensor([57, 0, 93]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u").cuda()),
               (torch.tensor([3, 99, 41]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u2").cuda()),
               (torch.tensor([12, 12, 70]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u3").cuda())]
    dt_e, dt_g = make(), make()
    eager = Solver(TrainStep(dt_e, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    graph = GraphSolver(TrainStep(dt_g, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    key = "transformer.blocks.1.mlp.0.weight"
    for it, (t, u) in enumerate(batches):
        eager.lr = graph.lr = 1e-3 * (it + 1)
        oe = eager.step(x0, cond, t, pt, u)
        og = graph.step(x0, cond, t, pt, u)
        le, lg, ne, ng = float(oe["loss"]), float(og["loss"]), float(oe["grad_norm"]), float(og["grad_norm"])
        print("iter %d: loss eager %.6f graph %.6f   |g| eager %.5f graph %.5f" % (it, le, lg, ne, ng))
        assert abs(le - lg) <= 1e-5 * abs(le) and abs(ne - ng) <= 1e-4 * ne
        # Adam's step is lr * m / (sqrt(v) + eps): for the handful of elements whose gradient is rounding noise around
        # zero it is +-lr whichever way the noise fell, so the weights are compared on the bulk, not on the maximum
        we, wg = dict(dt_e.named_parameters())[key].detach(), dict(dt_g.named_parameters())[key].detach()
        d = (wg - we).abs()
        assert d.mean().item() < 1e-7 and (d > 1e-6).float().mean().item() < 1e-4, (d.mean().item(), d.max().item())
    assert torch.equal(dt_e.Lt_count.cpu(), dt_g.Lt_count.cpu())
    assert graph.train_step.loss_scale_exp == eager.train_step.loss_scale_exp


def test_graph_solver_resumes_from_its_state_dict():
    """a run trained with GraphSolver can be checkpointed and resumed -- two iterations, state_dict + the
    weights into a fresh model / solver, a third iteration there == the third iteration of the uninterrupted run (loss, the
    AdamW moments inside the graph, the bias-correction counter), and the saturation monitor saw real gradients."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, GraphSolver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep

    def make():
        m = build_model(default_config(n_layer=2, diffusion_step=100))
        m.load_state_dict({**dict(synth_sd("dalle", 2)), **synth_sd("encoder")}, strict=False)
        m = m.cuda().eval()
        dt = m.transformer
        dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
        return dt
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0").cuda()
    cond = synth.synth_cond_emb(3, key="tl.c").cuda()
    pt = (torch.ones(3) / 100).cuda()
    batches = [(torch.tensor([57, 0, 93]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u").cuda()),
               (torch.tensor([3, 99, 41]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u2").cuda()),
               (torch.tensor([12, 12, 70]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u3").cuda())]
    dt_a, dt_b = make(), make()
    a = GraphSolver(TrainStep(dt_a, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    for t, u in batches[:2]:
        a.step(x0, cond, t, pt, u)
    assert float(a.train_step._amax_live) > 0.0            # the monitor accumulates max |scaled dY| on the device
    state = a.state_dict()
    weights = {k: v.detach().clone() for k, v in dt_a.state_dict().items()}
    oa = a.step(x0, cond, batches[2][0], pt, batches[2][1])
    dt_b.load_state_dict(weights)
    b = GraphSolver(TrainStep(dt_b, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    b.load_state_dict(state)
    ob = b.step(x0, cond, batches[2][0], pt, batches[2][1])
    assert b.iteration_graph.iteration == a.iteration_graph.iteration == 3 and b.clip_grad_norm.last_epoch == 2
    la, lb = float(oa["loss"]), float(ob["loss"])
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    key = "transformer.blocks.1.mlp.0.weight"
    ma, mb = a.iteration_graph.opt_state[key][0], b.iteration_graph.opt_state[key][0]
    assert (ma - mb).abs().max().item() <= 1e-5 * ma.abs().max().item()
    wa, wb = dict(dt_a.named_parameters())[key].detach(), dict(dt_b.named_parameters())[key].detach()
    d = (wa - wb).abs()
    assert d.mean().item() < 1e-7 and (d > 1e-6).float().mean().item() < 1e-4


def test_weight_swap_recaptures_before_the_next_replay():
    """weights replaced behind a captured iteration (checkpoint / EMA load -> TrainStep.reset_scales) must
    re-capture the graph BEFORE the next replay -- its pre-scales 2^s and loss scale belong to the old weights, and
    weights a few times larger would saturate the fp16 planes silently.  An 8x larger mlp.0 weight is swapped in: the step
    after the swap must equal a freshly built solver's step on the same weights (not one replay with stale constants)."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, GraphSolver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep

    def make():
        m = build_model(default_config(n_layer=2, diffusion_step=100))
        m.load_state_dict({**dict(synth_sd("dalle", 2)), **synth_sd("encoder")}, strict=False)
        m = m.cuda().eval()
        dt = m.transformer
        dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
        return dt
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0").cuda()
    cond = synth.synth_cond_emb(3, key="tl.c").cuda()
    pt = (torch.ones(3) / 100).cuda()
    t1, u1 = torch.tensor([57, 0, 93]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u").cuda()
    t2, u2 = torch.tensor([3, 99, 41]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u2").cuda()
    dt_a, dt_b = make(), make()
    a = GraphSolver(TrainStep(dt_a, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    a.step(x0, cond, t1, pt, u1)
    graph_before = a.iteration_graph.graph
    swapped = {k: v.detach().clone() for k, v in dt_b.state_dict().items()}
    key = "transformer.blocks.1.mlp.0.weight"
    swapped[key] = swapped[key] * 8.0
    dt_a.load_state_dict(swapped)
    dt_b.load_state_dict(swapped)
    a.train_step.reset_scales()                    # what solver._invalidate does on a weight swap
    oa = a.step(x0, cond, t2, pt, u2)
    assert a.iteration_graph.graph is not graph_before, "the stale graph was replayed"
    b = GraphSolver(TrainStep(dt_b, precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5))
    ob = b.step(x0, cond, t2, pt, u2)
    la, lb = float(oa["loss"]), float(ob["loss"])
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    na, nb = float(oa["grad_norm"]), float(ob["grad_norm"])
    assert abs(na - nb) <= 1e-4 * abs(nb), (na, nb)


def test_graph_solver_two_segments_with_reduction_hook():
    """The data-parallel form of the captured iteration (engine/solver_spec.py:109: DDP reduces before the optimizer step):
    GraphSolver(reduce=...) replays a gradients graph, calls reduce(grads) on the graph's own gradient tensors, replays the
    clip + AdamW graph.  With an identity reduction it must equal the one-graph solver to rounding over three iterations
    (one call of the hook per iteration); a reduction that halves the gradients must act on the update: with the clip
    active at the halved norm as well the step direction is the same and the reported norm is half."""
    from conftest import synth_sd
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, GraphSolver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep

    def make():
        m = build_model(default_config(n_layer=2, diffusion_step=100))
        m.load_state_dict({**dict(synth_sd("dalle", 2)), **synth_sd("encoder")}, strict=False)
        dt = m.cuda().eval().transformer
        dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
        return dt
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="tl.x0").cuda()
    cond = synth.synth_cond_emb(3, key="tl.c").cuda()
    pt = (torch.ones(3) / 100).cuda()
    batches = [(torch.tensor([57, 0, 93]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u").cuda()),
               (torch.tensor([3, 99, 41]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u2").cuda()),
               (torch.tensor([12, 12, 70]).cuda(), synth.synth_uniform((3, 257, 265), key="tl.u3").cuda())]
    calls = []

    def identity(grads):
        calls.append(len(grads))

    def halve(grads):
        torch._foreach_mul_(list(grads.values()), 0.5)
    dts = [make(), make(), make()]
    solvers = [GraphSolver(TrainStep(dts[0], precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5)),
               GraphSolver(TrainStep(dts[1], precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5),
                           reduce=identity),
               GraphSolver(TrainStep(dts[2], precision="f16x2"), lr=1e-3, clip_grad_norm=GradClipWindow(0, 5000, 0.5),
                           reduce=halve)]
    outs = [[s.step(x0, cond, t, pt, u) for t, u in batches] for s in solvers]
    assert solvers[0].iteration_graph.update_graph is None and solvers[1].iteration_graph.update_graph is not None
    assert calls == [63, 63, 63]
    for a, b in zip(outs[0], outs[1]):      # (two runs of the same iteration agree to rounding, not bit for bit: the loss
        la, lb, na, nb = float(a["loss"]), float(b["loss"]), float(a["grad_norm"]), float(b["grad_norm"])   # sums use atomics)
        assert abs(la - lb) <= 1e-6 * abs(la) and abs(na - nb) <= 1e-5 * abs(na), (la, lb, na, nb)
    pa, pb = dict(dts[0].named_parameters()), dict(dts[1].named_parameters())
    for k in pa:
        if k.endswith("key.bias"):      # softmax is invariant to a key bias: its gradient is rounding noise, which AdamW
            continue                    # normalises to full-size steps -- not comparable between two runs
        d = (pa[k].detach() - pb[k].detach()).abs()
        assert d.mean().item() < 1e-6 and (d > 1e-4).float().mean().item() < 1e-4, (k, d.mean().item(), d.max().item())
    # halved gradients: same loss at the first iteration, half the norm; the norms (~1e2) are far above the clip (0.5) in
    # both runs, so the clipped update is the same direction and size up to rounding
    assert abs(float(outs[2][0]["loss"]) - float(outs[0][0]["loss"])) <= 1e-6 * abs(float(outs[0][0]["loss"]))
    r = float(outs[2][0]["grad_norm"]) / float(outs[0][0]["grad_norm"])
    assert abs(r - 0.5) < 1e-5, r
    assert float(outs[0][0]["grad_norm"]) > 2.0        # (so that both runs clip)


def test_ema_multi_matches_the_reference_expression(L):
    """ds_ema_multi / solver.EMA on the GPU: ema = ema * decay + current * (1 - decay) over many tensors in one pass (engine/ema.py:
    40-56) -- bit for bit what the per-tensor torch expression gives, incl. odd sizes, unaligned views and more tensors than one
    launch carries; non-fp32 entries take the per-tensor path."""
    from text_to_sound_synthesis_amd.modeling.solver import EMA

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ws = torch.nn.ParameterList([torch.nn.Parameter(rnd((n,), "ema.w%d" % i)) for i, n in enumerate(
                [1, 3, 4097, 8192, 50001] + [257 + 13 * j for j in range(120)])])
            self.register_buffer("count", torch.arange(5))                      # int64: not for the kernel
            self.register_buffer("table", rnd((100, 37), "ema.t"))
    net = Net().cuda()
    ema = EMA(net, decay=0.99, update_interval=1, device="cuda")
    want = {k: v.clone() for k, v in ema.ema.items()}
    for it in range(3):
        with torch.no_grad():
            for p_ in net.parameters():
                p_.add_(0.01 * (it + 1))
            net.count += 1
        cur = net.state_dict()
        for k in want:
            want[k] = (want[k] * 0.99 + cur[k].detach() * (1 - 0.99)).to(want[k].dtype)
        ema.update(iteration=it)
    for k in want:
        bad = (ema.ema[k] != want[k])
        assert not bool(bad.any()), (k, int(bad.sum()), bad.numel(), float((ema.ema[k].double() - want[k].double()).abs().max()),
                                    bad.flatten().nonzero().flatten()[:8].tolist())


def test_grad_norm_multi_and_clip_coefficient(L):
    """ds_grad_norm_multi: the global L2 norm over many gradient tensors (odd sizes, an unaligned view, more tensors than one launch
    carries) and torch's clip coefficient min(1, max_norm / (norm + 1e-6)) written to device memory -- against float64."""
    import ctypes
    sizes = [1, 3, 4095, 4096, 4097, 50001, 1024 * 1024] + [300 + 7 * j for j in range(200)]
    ts = [rnd((n,), "gn.%d" % i, 0.05 + 0.01 * (i % 5)).cuda() for i, n in enumerate(sizes)]
    ts.append(rnd((1000,), "gn.view", 0.1).cuda()[1:])                      # 4-byte aligned only
    want = math.sqrt(sum(float(t.double().pow(2).sum()) for t in ts))
    rec = (ctypes.c_int64 * (2 * len(ts)))()
    chunks = 0
    for i, t in enumerate(ts):
        rec[2 * i:2 * i + 2] = [t.data_ptr(), t.numel()]
        chunks += (t.numel() + 4095) // 4096
    part = torch.full((chunks,), float("nan"), dtype=torch.float64, device="cuda")
    for max_norm in (0.5, 1e9, 0.0):
        total, coef = torch.zeros(1, device="cuda"), torch.full((1,), -1.0, device="cuda")
        L.check(L.lib().ds_grad_norm_multi(ctypes.cast(rec, ctypes.c_void_p), len(ts), L.ptr(part), chunks, max_norm, L.ptr(total),
                                           L.ptr(coef), L.stream()))
        assert abs(float(total) - want) < 1e-6 * want, (float(total), want)
        want_c = 1.0 if max_norm <= 0 else min(1.0, max_norm / (want + 1e-6))
        assert abs(float(coef) - want_c) < 1e-6 * want_c, (float(coef), want_c)
    ref = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(ts)))
    assert abs(float(total) - float(ref)) < 1e-5 * want
    total2 = torch.zeros(1, device="cuda")
    L.check(L.lib().ds_grad_norm_multi(ctypes.cast(rec, ctypes.c_void_p), len(ts), L.ptr(part), chunks, 0.5, L.ptr(total2), None, L.stream()))
    assert float(total2) == float(total)                                    # fixed-order sums: bit-reproducible, coef optional
