"""GPU parity of entry points added after the round's GPU budget was used up (so: written against goldens and the
CPU-verified oracle, first executed by the round-end run).  Sorted after the other GPU files on purpose.

  * DALLE.sample -- the trainer's logging sampler (dalle_spec.py:264-343) vs the reference's images
  * VQModel.forward / get_input (spec_codec/vqgan.py:72-82)
"""
import os

import pytest
import torch

from conftest import golden
from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
NO_GRAD = True          # tests/conftest.py: every test of this module runs under torch.no_grad()

MEL_TOL = 1e-3      # BASELINE.json north_star: max-abs on mel


def test_dalle_sample_logging_sampler_vs_reference():
    from test_hip_models import build
    g, ge = golden("dalle_sample_T10_L2"), golden("encoder_T10_L2")
    m = build(2, T=10)
    assert m.transformer.truncation_r is None and m.transformer.truncation_k is None     # no wrapper in this entry point
    mel = (synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1).cuda()
    cond = synth.synth_cond_emb(2, key="traj.cond").cuda()
    # the synthetic codebook has rounding-level ties in the nearest-code search (tested in test_hip_models.py): the
    # encoder runs, but the re-sampling starts from the reference's own tokens so that the images are comparable
    real_get_tokens = m.get_tokens
    own = {}

    def get_tokens(spec):
        qz, tok = real_get_tokens(spec)
        own["tokens"] = tok
        return qz, ge["tokens"].cuda()
    m.get_tokens = get_tokens
    n = [0]

    def noise(_, shp):
        n[0] += 1
        return synth.synth_uniform(shp, key="ds.u%d" % (n[0] - 1))
    out = m.sample({"image": mel, "text": ["a", "b"], "condition_embed_token": cond}, filter_ratio=[0, 0.5, 1.0],
                   content_ratio=[1], noise_fn=noise)
    assert n[0] == int(g["calls"]) == 27
    assert (own["tokens"].cpu() == ge["tokens"]).float().mean() > 0.7
    assert out["condition"] == ["a", "b"] and out["input_image"] is mel
    s = slice(None, None, int(g["time_stride"]))
    for key, name in (("reconstruction_image", "reconstruction"), ("cond1_cont1_fr0_image", "fr0"),
                      ("cond1_cont1_fr0.5_image", "fr05"), ("cond1_cont1_fr1.0_image", "fr1")):
        img = out[key].cpu()
        assert img.shape == (2, 1, 80, 848)
        err = (img[..., s] - g[name]).abs().max().item()
        print("%s: max-abs err vs reference %.2e" % (key, err))
        assert err < MEL_TOL, key
    assert m.training                                   # the reference leaves the model in train() mode (:339)
    with pytest.raises(NotImplementedError):
        m.sample({"image": mel, "condition_embed_token": cond}, return_att_weight=True)


def test_vqmodel_forward_is_decode_of_encode():
    from test_hip_models import build
    m = build(2, T=10)
    codec = m.content_codec
    mel = (synth.synth_uniform((2, 1, 80, 848), key="enc.mel") * 2 - 1).cuda()
    dec, diff = codec(mel)
    quant, loss, _ = codec.encode(mel)
    assert dec.shape == (2, 1, 80, 848) and torch.equal(dec, codec.decode(quant)) and torch.equal(diff, loss)
    x = codec.get_input({"image": mel[:, 0]}, "image")
    assert x.shape == (2, 1, 80, 848) and torch.equal(x, mel)


# ---- the per-sample ping-pong program of the split GEMM (gemm_f16x2_ps.hip; ds_gemm_f16x2_force_tile(9)) -----------
# The default launch takes it only for grids of whole CU rounds (B = 64: covered by the batch-size sweep and the
# full-configuration parity tests); force_tile(9) runs it on small batches so that every store family, every sample
# offset (265 b mod 16), the last sample's clamped row groups and 2 .. 128 k-tiles are compared bit for bit with the
# loader-split 4-wave GEMM.
PS = 9


@pytest.mark.parametrize("B,N,K", [(5, 1024, 1024), (3, 1024, 4096), (17, 256, 64), (2, 4096, 1024), (16, 512, 128)])
def test_f16x2_per_sample_program_bit_identical(B, N, K):
    """Row-major output with bias + residual (in place, like the denoiser's projections) and packed split output with
    GELU2: same bits as the loader-split 4-wave GEMM; rows of other samples inside a tile's 288-row window and the
    ninth block's rows past the sample are never written."""
    from test_hip_split_gemm import relerr, rnd, torch_split
    from text_to_sound_synthesis_amd import _lib as L
    Lq = 265
    M = B * Lq
    A, W, b, R = rnd((M, K), "ps.A", 2.0).cuda(), rnd((N, K), "ps.W", 0.05).cuda(), rnd((N,), "ps.b").cuda(), \
        rnd((M, N), "ps.R").cuda()
    W2, sc = L.split_f16x2(W)
    W2p, _ = L.split_f16x2(W, packed=True)
    A2p = L.pack_planes(torch_split(A))
    M16 = (M + 15) // 16 * 16
    ref = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, ref, M, N, K, bias=b, R=R, split2=sc)
    assert relerr((ref - R).cpu(), (A.double() @ W.double().t() + b.double()).float().cpu()) < 2e-6
    refg = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, refg, M, N, K, bias=b, act=L.ACT_GELU2, split2=sc)
    refn = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, refn, M, N, K, bias=b, split2=sc)
    try:
        L.lib().ds_gemm_f16x2_force_tile(PS)
        for rep in range(2):
            out = R.clone()                                   # in place: C aliases R
            L.gemm(A2p, W2p, out, M, N, K, bias=b, R=out, split2=sc, a_plane=M16 * K, rows_per_sample=Lq)
            assert torch.equal(out, ref)
        guard = torch.full((M + 64, N), float("nan"), device="cuda")      # nothing past row M is touched
        L.gemm(A2p, W2p, guard, M, N, K, bias=b, split2=sc, a_plane=M16 * K, rows_per_sample=Lq)
        assert torch.isnan(guard[M:]).all() and torch.equal(guard[:M], refn)
        outs = torch.zeros(2, M16 * N, device="cuda", dtype=torch.float16)
        L.gemm(A2p, W2p, outs, M, N, K, bias=b, act=L.ACT_GELU2, split2=sc, a_plane=M16 * K, c_plane=M16 * N,
               rows_per_sample=Lq)
        assert torch.equal(L.unpack_planes(outs, M, N), torch_split(refg))
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)


@pytest.mark.parametrize("B,N,K", [(20, 4096, 128), (35, 4096, 64), (17, 4096, 256), (70, 1024, 64)])
def test_f16x2_leading_samples_on_the_per_sample_program(B, N, K):
    """With rows_per_sample given, a batch whose per-sample tiles do not fill whole rounds of the chip (20 samples x 16 column
    tiles = 1.25 rounds) runs its leading whole rounds (16 samples) on the per-sample program and the rest as a second, smaller
    launch: the same bits as one launch of a 4-wave tile, with bias and an in-place residual, nothing written past row M."""
    from test_hip_split_gemm import rnd, torch_split
    from text_to_sound_synthesis_amd import _lib as L
    Lq = 265
    M = B * Lq
    A, W, b, R = rnd((M, K), "ls.A", 2.0).cuda(), rnd((N, K), "ls.W", 0.05).cuda(), rnd((N,), "ls.b").cuda(), rnd((M, N), "ls.R").cuda()
    W2p, sc = L.split_f16x2(W, packed=True)
    A2p = L.pack_planes(torch_split(A))
    M16 = (M + 15) // 16 * 16
    L.lib().ds_gemm_f16x2_force_tile(1)
    try:
        ref = R.clone()
        L.gemm(A2p, W2p, ref, M, N, K, bias=b, R=ref, split2=sc, a_plane=M16 * K)
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)
    out = torch.full((M + 40, N), float("nan"), device="cuda")
    out[:M] = R
    L.gemm(A2p, W2p, out, M, N, K, bias=b, R=out, split2=sc, a_plane=M16 * K, rows_per_sample=Lq)
    assert torch.equal(out[:M], ref) and torch.isnan(out[M:]).all()


@pytest.mark.parametrize("B", [1, 3, 8])
def test_f16x2_per_sample_program_attention_store_bit_identical(B):
    """The attention-ready stores (QKV: Q planes, K image, V^T image; cross-attention Q alone) through the per-sample
    tiles: units of 8 keys that straddle the 128 / 256-row slab edges are written in two parts."""
    from test_hip_split_gemm import rnd, torch_split
    from text_to_sound_synthesis_amd import _lib as L
    Lq, H, D = 265, 16, 1024
    M, K = B * Lq, D
    A = rnd((M, K), "pa.A", 2.0).cuda()
    A2p = L.pack_planes(torch_split(A))
    M16 = (M + 15) // 16 * 16
    for N in (3 * D, D):
        W, b = rnd((N, K), "pa.W%d" % N, 0.05).cuda(), rnd((N,), "pa.b%d" % N).cuda()
        W2p, sc = L.split_f16x2(W, packed=True)

        def run():
            qh = torch.full((2, B, H, Lq, 64), float("nan"), device="cuda", dtype=torch.float16)
            img = torch.zeros(B, H, 4, 288 * 64, device="cuda", dtype=torch.float16) if N == 3 * D else None
            L.gemm(A2p, W2p, qh, M, N, K, bias=b, split2=sc, a_plane=M16 * K, store=L.STORE_ATTN, rows_per_sample=Lq,
                   attn=(img, H, 288, B * H * Lq * 64))
            return qh, img
        L.lib().ds_gemm_f16x2_force_tile(2 if B == 1 else 1)   # 4-wave tiles (checked against the host packing elsewhere)
        try:
            q_ref, img_ref = run()
            L.lib().ds_gemm_f16x2_force_tile(PS)
            qh, img = run()
        finally:
            L.lib().ds_gemm_f16x2_force_tile(-1)
        assert torch.equal(qh, q_ref)
        assert img is None or torch.equal(img, img_ref)


HALF = 10          # ds_gemm_f16x2_force_tile(10): the half-tile program (a 272-row sample = a 144-row + a 128-row tile)


def _exact_rows(M, Lp, prog):
    """Rows whose bits equal the 4-wave programs': all but the 16-row block on the 16x16x32 MFMA (rows 256..271 of a
    sample for full tiles, rows 128..143 for half tiles)."""
    r = torch.arange(M, device="cuda") % Lp
    return r < 256 if prog == PS else (r < 128) | (r >= 144)


@pytest.mark.parametrize("prog", [PS, HALF])
@pytest.mark.parametrize("B,N,K", [(5, 1024, 1024), (3, 1024, 4096), (2, 4096, 1024), (7, 256, 128), (4, 512, 64)])
def test_f16x2_per_sample_program_272_row_samples(B, N, K, prog):
    """Samples of 272 rows (17 packed row groups: the denoiser's padded-row mode): the tile's ninth block row is the 16
    rows 256..271 on v_mfma_f32_16x16x32_f16.  Rows 0..255 of every sample keep the bits of the 4-wave programs; rows
    256..271 sum the same products in another order (one 32-k MFMA instead of two 16-k ones) and are compared with
    float64.  Row-major + bias + residual in place, and packed split output with GELU2.  prog = HALF: the same through the
    half-tile program (three LDS stages, two phases per k-tile; k-tile counts 2, 4, 32 and 128 cover every remainder of its
    3-k-tile loop): there the 16-row block is rows 128..143 of a sample."""
    from test_hip_split_gemm import relerr, rnd, torch_split
    from text_to_sound_synthesis_amd import _lib as L
    Lp = 272
    M = B * Lp
    A, W, b, R = rnd((M, K), "p272.A", 2.0).cuda(), rnd((N, K), "p272.W", 0.05).cuda(), rnd((N,), "p272.b").cuda(), \
        rnd((M, N), "p272.R").cuda()
    W2, sc = L.split_f16x2(W)
    W2p, _ = L.split_f16x2(W, packed=True)
    A2p = L.pack_planes(torch_split(A))
    ref = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, ref, M, N, K, bias=b, R=R, split2=sc)
    refg = torch.empty(M, N, device="cuda")
    L.gemm(A, W2, refg, M, N, K, bias=b, act=L.ACT_GELU2, split2=sc)
    exact = (A.double() @ W.double().t() + b.double())
    low = _exact_rows(M, Lp, prog)                               # rows computed by the 32x32x16 blocks
    try:
        L.lib().ds_gemm_f16x2_force_tile(prog)
        out = R.clone()
        L.gemm(A2p, W2p, out, M, N, K, bias=b, R=out, split2=sc, a_plane=M * K, rows_per_sample=Lp)
        assert torch.equal(out[low], ref[low])
        assert relerr((out - R)[~low].cpu(), exact[~low].float().cpu()) < 2e-6
        guard = torch.full((M + 64, N), float("nan"), device="cuda")
        L.gemm(A2p, W2p, guard, M, N, K, bias=b, split2=sc, a_plane=M * K, rows_per_sample=Lp)
        assert torch.isnan(guard[M:]).all() and not torch.isnan(guard[:M]).any()
        outs = torch.zeros(2, M * N, device="cuda", dtype=torch.float16)
        L.gemm(A2p, W2p, outs, M, N, K, bias=b, act=L.ACT_GELU2, split2=sc, a_plane=M * K, c_plane=M * N,
               rows_per_sample=Lp)
        got = L.unpack_planes(outs, M, N)
        want = torch_split(refg)
        assert torch.equal(got[:, low], want[:, low])
        gv = got[0].float() + got[1].float()
        eg = exact * torch.sigmoid(1.702 * exact)
        assert relerr(gv[~low].cpu(), eg[~low].float().cpu()) < 4e-6
    finally:
        L.lib().ds_gemm_f16x2_force_tile(-1)


@pytest.mark.parametrize("prog", [PS, HALF])
@pytest.mark.parametrize("B", [1, 4])
def test_f16x2_per_sample_program_272_row_attention_store(B, prog):
    """The attention-ready stores for 272-row samples: Q planes [B][heads][272][64], K / V^T images with 272 keys (the
    denoiser masks keys 265.. in the attention kernel).  Positions 0..255 bit-identical to the 4-wave programs, 256..271
    compared as values."""
    from test_hip_split_gemm import rnd, torch_split
    from text_to_sound_synthesis_amd import _lib as L
    Lp, H, D = 272, 16, 1024
    M, K = B * Lp, D
    A = rnd((M, K), "pa272.A", 2.0).cuda()
    A2p = L.pack_planes(torch_split(A))
    for N in (3 * D, D):
        W, b = rnd((N, K), "pa272.W%d" % N, 0.05).cuda(), rnd((N,), "pa272.b%d" % N).cuda()
        W2p, sc = L.split_f16x2(W, packed=True)

        def run():
            qh = torch.full((2, B, H, Lp, 64), float("nan"), device="cuda", dtype=torch.float16)
            img = torch.zeros(B, H, 4, 288 * 64, device="cuda", dtype=torch.float16) if N == 3 * D else None
            L.gemm(A2p, W2p, qh, M, N, K, bias=b, split2=sc, a_plane=M * K, store=L.STORE_ATTN, rows_per_sample=Lp,
                   attn=(img, H, 288, B * H * Lp * 64))
            return qh, img
        L.lib().ds_gemm_f16x2_force_tile(2 if B == 1 else 1)
        try:
            q_ref, img_ref = run()
            L.lib().ds_gemm_f16x2_force_tile(prog)
            qh, img = run()
        finally:
            L.lib().ds_gemm_f16x2_force_tile(-1)
        keep = torch.arange(Lp, device="cuda")
        keep = keep < 256 if prog == PS else (keep < 128) | (keep >= 144)       # positions outside the 16-row block
        assert torch.equal(qh[:, :, :, keep], q_ref[:, :, :, keep])
        val = lambda x: x[0].float() + x[1].float()
        assert not torch.isnan(qh.float()).any()
        assert (val(qh) - val(q_ref)).abs().max().item() < 2e-5 * val(q_ref).abs().max().item()
        if img is not None:
            kk = lambda im: im[:, :, 0].float() + im[:, :, 1].float()      # K image: hi + lo, [B][H][288*64]
            vv = lambda im: im[:, :, 2].float() + im[:, :, 3].float()
            assert (kk(img) - kk(img_ref)).abs().max().item() < 2e-5 * kk(img_ref).abs().max().item()
            assert (vv(img) - vv(img_ref)).abs().max().item() < 2e-5 * vv(img_ref).abs().max().item()
            # key slots 272.. stay zero; most of the image (keys outside the 16-row block) is bit-identical
            same = (img == img_ref).float().mean().item()
            assert same > 0.9, same


def test_denoiser_step_with_per_sample_program_gives_the_same_tokens():
    """A whole sampling step of the 19-layer denoiser at B = 8 with every eligible GEMM forced onto the per-sample
    program: logits and tokens equal the 4-wave programs'."""
    from test_hip_split_gemm import build
    from text_to_sound_synthesis_amd import _lib as L
    m = build(19, mode="f16x2")
    dt = m.transformer
    dt.truncation_r = 0.85
    B = 8
    cond = synth.synth_cond_emb(B, key="bt.cond").cuda()
    x = synth.synth_tokens(B, mask_frac=0.6, key="bt.x").cuda()
    t = torch.full((B,), 41, device="cuda", dtype=torch.long)
    u = synth.synth_uniform((B, 257, 265), key="bt.u").cuda()
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    want = dt.p_sample_tokens(x, kv, t, u, False).clone()
    want_logits = dt.transformer(x, cond, t).clone()
    try:
        L.lib().ds_gemm_f16x2_force_tile(PS)
        # the driver sizes its rows with the same rule that dispatches the program (ds_gemm_f16x2_ps_taken, forced tile
        # included): forced, the step runs in padded-row mode unless that is switched off
        assert L.lib().ds_denoiser_rows_per_sample(dt.transformer.packed(dt._schedule_table())["handle"], B) == 272
        dt.transformer.row_padding = False
        got = dt.p_sample_tokens(x, kv, t, u, False)
        assert torch.equal(got, want)                                  # 265-row tiles: the bits of the 4-wave programs
        assert torch.equal(dt.transformer(x, cond, t), want_logits)
        dt.transformer.row_padding = True
        got = dt.p_sample_tokens(x, kv, t, u, False)                   # 272-row tiles: rows 256..264 summed in another order
        assert int((got != want).sum()) <= 2
    finally:
        dt.transformer.row_padding = True
        L.lib().ds_gemm_f16x2_force_tile(-1)


def test_padded_row_mode_of_the_sampling_step():
    """B = 64, 19 layers: the sampling step in padded-row mode (272 rows per sample, the per-sample GEMM program with a
    16-row ninth block, 7 zero-embedded extra rows per sample that are queries but never keys) against the same step on
    265 rows: the same tokens up to exact near-ties, from a mixed state and from the all-[MASK] start; then five chained
    steps.  (Rows 256..264 of a sample are summed in another order: logits agree to ~1e-6, not bit for bit.)"""
    from test_hip_split_gemm import build
    m = build(19, mode="f16x2")
    dt = m.transformer
    dt.truncation_r = 0.85
    tr = dt.transformer
    B = 64
    cond = synth.synth_cond_emb(B, key="pad.cond").cuda()
    kv = tr.condition_kv(cond, dt._schedule_table())
    u = synth.synth_uniform((B, 257, 265), key="pad.u").cuda()

    def step(x, t, initial, pad):
        tr.row_padding = pad
        return dt.p_sample_tokens(x, kv, torch.full((B,), t, device="cuda", dtype=torch.long), u, initial).clone()
    try:
        x = synth.synth_tokens(B, mask_frac=0.6, key="pad.x").cuda()
        a, b = step(x, 41, False, True), step(x, 41, False, False)
        diff = int((a != b).sum())
        print("mixed state, t = 41: %d of %d tokens differ between 272- and 265-row steps" % (diff, a.numel()))
        assert diff <= 2
        x0 = torch.full((B, 265), 256, dtype=torch.long, device="cuda")
        a, b = step(x0, 99, True, True), step(x0, 99, True, False)
        assert int((a != b).sum()) <= 2
        xa = xb = x0
        for i, t in enumerate((99, 98, 97, 96, 95)):
            xa, xb = step(xa, t, i == 0, True), step(xb, t, i == 0, False)
        same_clips = int((xa == xb).all(dim=1).sum())
        print("five chained steps: %d of %d clips identical" % (same_clips, B))
        assert same_clips >= B - 2 and int((xa == 256).sum()) == int((xb == 256).sum()) or same_clips >= B - 2
        assert int(xa.max()) <= 256 and int(xa.min()) >= 0
    finally:
        tr.row_padding = True


def test_batch_32_sampling_step_on_half_tiles():
    """BASELINE configs[1]'s batch: at B = 32 the N = 1024 GEMMs are 128 full tiles (half the CUs) and the QKV GEMM 1.5
    rounds, so the dispatch takes the HALF-tile program there (256 / 768 tiles) and full tiles for FC1 (512), all in
    padded-row mode.  The step must give the tokens of the 265-row step (4-wave programs for the GEMMs full tiles cannot
    fill) up to exact near-ties, from a mixed state, from the all-[MASK] start and over five chained steps."""
    from test_hip_split_gemm import build
    from text_to_sound_synthesis_amd import _lib as L
    m = build(19, mode="f16x2")
    dt = m.transformer
    dt.truncation_r = 0.85
    tr = dt.transformer
    B = 32
    cond = synth.synth_cond_emb(B, key="b32.cond").cuda()
    kv = tr.condition_kv(cond, dt._schedule_table())
    u = synth.synth_uniform((B, 257, 265), key="b32.u").cuda()
    h = tr.packed(dt._schedule_table())["handle"]
    assert L.lib().ds_denoiser_rows_per_sample(h, B) == 272 and L.lib().ds_denoiser_rows_per_sample(h, 16) == 265

    def step(x, t, initial, pad):
        tr.row_padding = pad
        return dt.p_sample_tokens(x, kv, torch.full((B,), t, device="cuda", dtype=torch.long), u, initial).clone()
    try:
        x = synth.synth_tokens(B, mask_frac=0.6, key="b32.x").cuda()
        a, b = step(x, 41, False, True), step(x, 41, False, False)
        diff = int((a != b).sum())
        print("B = 32, mixed state, t = 41: %d of %d tokens differ between the half-tile (272-row) and the 265-row step"
              % (diff, a.numel()))
        assert diff <= 2
        x0 = torch.full((B, 265), 256, dtype=torch.long, device="cuda")
        xa = xb = x0
        for i, t in enumerate((99, 98, 97, 96, 95)):
            xa, xb = step(xa, t, i == 0, True), step(xb, t, i == 0, False)
        same_clips = int((xa == xb).all(dim=1).sum())
        print("B = 32, five chained steps: %d of %d clips identical" % (same_clips, B))
        assert same_clips >= B - 2 and int(xa.max()) <= 256 and int(xa.min()) >= 0
    finally:
        tr.row_padding = True


# ---- loss settings the golden vector does not cover (other mask weights / auxiliary weights / timesteps) --------------
@pytest.mark.parametrize("aux,adaptive,mask_w,ts", [
    (5.0e-4, True, [1, 1], [99, 0, 1]),
    (0.0, True, [1, 1], [0, 0, 0]),
    (1.0e-2, False, [2.0, 0.5], [50, 7, 98]),
])
def test_training_loss_and_gradients_other_settings(aux, adaptive, mask_w, ts):
    """Loss value (forward path and TrainStep) and every gradient of the 2-layer model against autograd through the
    oracle -- which tests/test_oracle_live_reference.py pins to the reference for exactly these settings -- with
    non-uniform pt, non-unit mask weights, a constant auxiliary weight, no auxiliary term, and t in {0, 1, 99}."""
    import diffsound_oracle as O
    from conftest import synth_sd
    from test_hip_models import build
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    m = build(2, T=100)
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = aux, adaptive, mask_w
    x0 = synth.synth_tokens(3, mask_frac=0.0, key="lv.x0")
    cond = synth.synth_cond_emb(3, key="lv.c")
    t, pt = torch.tensor(ts), torch.tensor([0.01, 0.02, 0.005])
    u = synth.synth_uniform((3, 257, 265), key="lv.u")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k.startswith("transformer.transformer.") else v)
          for k, v in synth_sd("dalle", 2).items()}
    with torch.enable_grad():
        _, _, want, _ = O.train_loss(sd, x0, cond, t, pt, u, mask_weight=tuple(mask_w), auxiliary_loss_weight=aux,
                                     adaptive_auxiliary_loss=adaptive)
        want.backward()
    dt.sample_time = lambda b, device, method="uniform": (t.cuda(), pt.cuda())
    out = dt({"content_token": x0.cuda(), "condition_embed_token": cond.cuda()}, return_loss=True, noise=u)
    assert abs(out["loss"].item() - want.item()) < 2e-4 * abs(want.item())
    with torch.no_grad():
        loss, grads = TrainStep(dt).loss_and_grads(x0.cuda(), cond.cuda(), t.cuda(), pt.cuda(), u.cuda())
    assert abs(loss.item() - want.item()) < 2e-4 * abs(want.item())
    worst = 0.0
    for k, v in sd.items():
        if not (k.startswith("transformer.transformer.") and v.is_floating_point() and v.grad is not None):
            continue
        mag = v.grad.abs().max().item()
        if mag < 1e-7:
            continue
        got = grads[k[len("transformer."):]].cpu().double()
        worst = max(worst, (got - v.grad.double()).abs().max().item() / mag)
    print("worst gradient error %.2e" % worst)
    assert worst < 2e-3


@pytest.mark.parametrize("B,H,W,Cout", [(2, 80, 848, 128), (1, 7, 5, 64), (3, 16, 33, 256)])
def test_conv3x3_one_input_channel_direct(B, H, W, Cout):
    """ds_conv3x3_c1 = Encoder.conv_in (diffusionmodules/model.py:423-427, :480): Conv2d(1, Cout, 3, padding 1) on the one-channel mel,
    channels-last output -- against torch's conv2d in float64, borders (zero padding) included."""
    from text_to_sound_synthesis_amd import _lib as L
    x = synth.synth_uniform((B, 1, H, W), key="c1.x%d" % H) * 2 - 1
    w = (synth.synth_uniform((Cout, 1, 3, 3), key="c1.w%d" % Cout) * 2 - 1) * 0.3
    b = (synth.synth_uniform((Cout,), key="c1.b%d" % Cout) * 2 - 1) * 0.1
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    xc, wc, bc = x[:, 0].contiguous().cuda(), w.reshape(Cout, 9).contiguous().cuda(), b.cuda()
    out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    nch = L.lib().ds_conv3x3_c1_chunks(H, W)
    part = torch.full((B, nch, 2, Cout), float("nan"), device="cuda", dtype=torch.float64)
    L.check(L.lib().ds_conv3x3_c1(L.ptr(xc), L.ptr(wc), L.ptr(bc), L.ptr(out), B, H, W, Cout, L.ptr(part), L.stream()))
    err = float((out.cpu().double() - want).abs().max())
    assert err < 2e-6, err
    # the GroupNorm partial sums of the output (per row segment): summed over the chunks they are the per-channel sums
    sums = part.sum(1).cpu()
    assert float((sums[:, 0] - want.sum((1, 2))).abs().max()) < 1e-4 * float(want.abs().sum((1, 2)).max())
    assert float((sums[:, 1] - want.pow(2).sum((1, 2))).abs().max()) < 1e-5 * float(want.pow(2).sum((1, 2)).max())
    out2 = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    L.check(L.lib().ds_conv3x3_c1(L.ptr(xc), L.ptr(wc), L.ptr(bc), L.ptr(out2), B, H, W, Cout, None, L.stream()))
    assert torch.equal(out, out2)
    # a row wider than the kernel's LDS image, or a channel count it cannot split over a workgroup, is refused (not mis-computed)
    assert L.lib().ds_conv3x3_c1(L.ptr(xc), L.ptr(wc), L.ptr(bc), L.ptr(out2), B, H, 2047, Cout, None, L.stream()) != 0
    assert L.lib().ds_conv3x3_c1(L.ptr(xc), L.ptr(wc), L.ptr(bc), L.ptr(out2), B, H, W, 12, None, L.stream()) != 0
