"""N1 -- end-to-end same-seed parity AT THE BENCHMARKED CONFIGURATION (19 layers, T = 100, K = 256, top0.85r, 8 captions)
against tests/golden/traj_T100_L19.npz, which oracle/make_golden.py: traj_full() produced by running the reference's own
loop (diffusion_transformer.py:587-659, 100 x p_sample :639-641), dalle_spec.py:80-91 decode_to_img and
vocoder/modules.py:129 with the per-step noise injected.  GPU only.

The sampler is a chain of 100 x 8 x 265 = 212 000 discrete decisions (top-r cut, then Gumbel-argmax); a decision can
only come out differently where the reference itself sat on a near-tie.  The golden therefore carries, per decision, the
Gumbel-argmax margin `gap` and the top-r cut margin `tmargin`, and the tests demand:
  teacher-forced  (x_t of every step taken from the reference): every disagreeing token sits on a near-tie
                  (tmargin < CUT_TIE or gap < GAP_TIE), and there are at most MAX_FLIPS of them in 212 000 decisions;
  free-running    (the product's own chain): clips whose final tokens equal the reference's are counted (floor
                  MIN_EXACT_CLIPS of 8) and, for those, mel <= 1e-3 max-abs and waveform <= 1e-4 RMS (north_star);
                  a clip that leaves the reference's trajectory must do so at a near-tie decision.
Both precision modes run: `fp32` (v_mfma_f32_32x32x2_f32, the strict mode) and `f16x2` (the default, 3-pass fp16 split).
The measured counts are printed and written to gpurun_out/n1_parity_<mode>.json."""
import json
import os

import pytest
import torch

from conftest import ROOT, golden, parity_line, synth_sd
from text_to_sound_synthesis_amd import synth

pytestmark = pytest.mark.gpu
NO_GRAD = True          # tests/conftest.py: every test of this module runs under torch.no_grad()

MEL_TOL = 1e-3           # BASELINE.json north_star: max-abs on mel
WAVE_RMS_TOL = 1e-4      # BASELINE.json north_star: RMS on waveform
CUT_TIE = 2e-5           # |mass ranked before a class - r| below this: the top-r cut is a rounding-level tie
GAP_TIE = 2e-4           # Gumbel-argmax margin below this: the argmax is a rounding-level tie
# Gates sit at what has been measured (rounds 1-4, every box: 0 disagreements, 8 of 8 clips) plus the smallest allowance a
# different box's rounding could need; round 3 had 8 / 6 here, which a real regression would have passed.
MAX_FLIPS = {"fp32": 0, "f16x2": 2}   # teacher-forced disagreements allowed in 212 000 decisions (each must be a near-tie)
MIN_EXACT_CLIPS = 7      # free-running: clips (of 8) whose final 265 tokens must equal the reference's; a diverging clip
                         # must leave the reference's trajectory at a near-tie, which is printed


@pytest.fixture(scope="module")
def g():
    return golden("traj_T100_L19")


@pytest.fixture(scope="module")
def model():
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=19, diffusion_step=100))
    sd = dict(synth_sd("dalle", 19))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m = m.cuda().eval()
    m.transformer.truncation_r = 0.85
    return m


@pytest.fixture(scope="module")
def voc():
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    v = Generator(80, 32, 3)
    v.load_state_dict(synth_sd("generator"))
    return v.cuda().eval()


def noise(step, shape):
    return synth.synth_uniform(shape, key="n1.u%d" % step)


def set_precision(model, mode):
    tr = model.transformer.transformer
    tr.precision = mode
    tr.invalidate()          # repack for this GEMM mode


def report(mode, name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "n1_parity_%s.json" % mode)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[name] = payload
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print("N1 %s %s: %s" % (mode, name, json.dumps(payload)))
    brief = {k: v for k, v in payload.items() if k not in ("detail",) and not (k == "first_divergence" and not v)}
    parity_line("N1 %s %s: %s" % (mode, name, json.dumps(brief)))


def full_wave_rms(g, wave):
    """Round 6: the chain goldens carry `wave_full` -- clips `wave_full_clips` (0 and 7) at all 217 088 samples, fp32 --, so
    the MelGAN output of a SAMPLED clip is compared over its whole length, reflect-padded tail included
    (vocoder/modules.py:95-130), not only over the 32 768-sample heads.  wave: f32[B,1,217088] on the device, from the
    reference's tokens.  Returns (rms over the whole clip, rms over the last 32 768 samples), the worse clip of the two."""
    clips = [int(c) for c in g["wave_full_clips"]]
    ours = wave[clips, 0].cpu()
    ref = g["wave_full"]
    assert ours.shape == ref.shape == (len(clips), 217088)
    whole = (ours - ref).pow(2).mean(1).sqrt().max().item()
    tail = (ours[:, -32768:] - ref[:, -32768:]).pow(2).mean(1).sqrt().max().item()
    return whole, tail


def near_tie(g, step_idx, clip, pos):
    return float(g["tmargin"][step_idx, clip, pos]) < CUT_TIE or float(g["gap"][step_idx, clip, pos]) < GAP_TIE


@pytest.mark.parametrize("mode", ["fp32", "f16x2"])
def test_teacher_forced_100_steps_19_layers(model, g, mode):
    set_precision(model, mode)
    dt = model.transformer
    cond = g["cond_emb"].float().cuda()
    trace = g["step_tokens"].long()                      # [100, 8, 265]: tokens after the step at t = 99 - i
    B = trace.shape[1]
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips = []
    for i in range(100):
        t = 99 - i
        x_t = torch.full((B, 265), 256, dtype=torch.long) if i == 0 else trace[i - 1]
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(),
                                 noise(t, (B, 257, 265)).cuda(), initial=(i == 0)).cpu()
        for b, p in (tok != trace[i]).nonzero().tolist():
            flips.append({"t": t, "clip": b, "pos": p, "ours": int(tok[b, p]), "ref": int(trace[i, b, p]),
                          "gap": float(g["gap"][i, b, p]), "tmargin": float(g["tmargin"][i, b, p])})
    report(mode, "teacher_forced", {"decisions": 100 * B * 265, "flips": len(flips), "detail": flips[:16]})
    unexplained = [f for f in flips if not (f["tmargin"] < CUT_TIE or f["gap"] < GAP_TIE)]
    assert not unexplained, "token disagreements away from any near-tie: %s" % unexplained[:4]
    # fp32 = an exact FMA chain per element: 0 allowed; the near-tie allowance is for the split arithmetic only
    assert len(flips) <= MAX_FLIPS[mode], "%d teacher-forced disagreements in %d decisions: %s" % (len(flips), 100 * B * 265, flips[:4])


def test_teacher_forced_batch64_padded_rows(model, g):
    """The same 100 teacher-forced steps AT THE BENCHMARKED BATCH SIZE, where the step runs in padded-row mode (272 rows
    per sample, per-sample GEMM program with a 16-row ninth block: csrc/api.hip rows_per_sample): the 8 reference captions
    replicated 8 times = 64 clips.  Every replica must reproduce the reference's tokens (disagreements only at the
    reference's own near-ties, at most MAX_FLIPS["f16x2"] per replica set), and the 8 replicas of a caption must agree."""
    set_precision(model, "f16x2")
    dt = model.transformer
    assert dt.transformer.row_padding
    R = 8
    cond = g["cond_emb"].float().cuda().repeat(R, 1, 1)
    trace = g["step_tokens"].long()
    B0 = trace.shape[1]
    B = R * B0
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips, replica_mismatch = [], 0
    for i in range(100):
        t = 99 - i
        x_t = (torch.full((B0, 265), 256, dtype=torch.long) if i == 0 else trace[i - 1]).repeat(R, 1)
        u = noise(t, (B0, 257, 265)).repeat(R, 1, 1)
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(), u.cuda(), initial=(i == 0)).cpu()
        replica_mismatch += int((tok.view(R, B0, 265) != tok.view(R, B0, 265)[:1]).sum())
        for b, p in (tok != trace[i].repeat(R, 1)).nonzero().tolist():
            flips.append({"t": t, "clip": b % B0, "replica": b // B0, "pos": p, "gap": float(g["gap"][i, b % B0, p]),
                          "tmargin": float(g["tmargin"][i, b % B0, p])})
    report("f16x2", "teacher_forced_batch64_padded_rows", {"decisions": 100 * B * 265, "flips": len(flips),
                                                           "replica_mismatches": replica_mismatch, "detail": flips[:16]})
    unexplained = [f for f in flips if not (f["tmargin"] < CUT_TIE or f["gap"] < GAP_TIE)]
    assert not unexplained, "token disagreements away from any near-tie: %s" % unexplained[:4]
    assert len(flips) <= MAX_FLIPS["f16x2"] * R and replica_mismatch == 0


@pytest.mark.parametrize("mode", ["fp32", "f16x2"])
def test_free_running_tokens_mel_wave(model, voc, g, mode):
    set_precision(model, mode)
    dt = model.transformer
    cond = g["cond_emb"].float().cuda()
    trace = g["step_tokens"].long()
    B = trace.shape[1]
    steps = []

    def traced_noise(t, shape):
        steps.append(t)
        return noise(t, shape)
    # the product's own loop; its per-step tokens are re-derived below from a second, instrumented pass
    out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0, noise_fn=traced_noise)
    assert steps == list(range(99, -1, -1))
    tokens = out["content_token"].cpu()
    same = [bool(torch.equal(tokens[b], g["tokens"][b].long())) for b in range(B)]
    # where does a diverging clip leave the reference's trajectory?  (instrumented chain, same arithmetic)
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    x = torch.full((B, 265), 256, dtype=torch.long).cuda()
    first = {}
    for i in range(100):
        t = 99 - i
        x = dt.p_sample_tokens(x, kv, torch.full((B,), t, dtype=torch.long).cuda(), noise(t, (B, 257, 265)).cuda(),
                               initial=(i == 0))
        xc = x.cpu()
        for b in range(B):
            if b not in first and not torch.equal(xc[b], trace[i, b]):
                pos = (xc[b] != trace[i, b]).nonzero().flatten().tolist()
                first[b] = {"t": t, "pos": pos[:8], "near_tie": all(near_tie(g, i, b, p) for p in pos)}
    assert torch.equal(x.cpu(), tokens), "sample() and the step-by-step chain disagree"
    assert sorted(first) == [b for b in range(B) if not same[b]] or all(b in first for b in range(B) if not same[b])
    mel = model.decode_to_img(g["tokens"].long().cuda(), (B, 256, 5, 53))      # reference tokens -> mel (all 8 clips)
    mel_err = (mel[:, 0].cpu() - g["mel"]).abs().amax(dim=(1, 2))
    wave = voc(mel[:, 0], scale=0.5, shift=0.5)
    n = g["wave_head"].shape[1]
    wave_rms = (wave[:, 0, :n].cpu() - g["wave_head"]).pow(2).mean(1).sqrt()
    full_rms, tail_rms = full_wave_rms(g, wave)
    # end to end on the product's OWN tokens, for the clips that stayed on the reference's trajectory
    mel_own = model.decode_to_img(tokens.cuda(), (B, 256, 5, 53))
    wave_own = voc(mel_own[:, 0], scale=0.5, shift=0.5)
    e2e_mel = [(mel_own[b, 0].cpu() - g["mel"][b]).abs().max().item() for b in range(B)]
    e2e_rms = [(wave_own[b, 0, :n].cpu() - g["wave_head"][b]).pow(2).mean().sqrt().item() for b in range(B)]
    report(mode, "free_running", {
        "clips": B, "clips_with_identical_tokens": sum(same),
        "token_agreement": float((tokens == g["tokens"].long()).float().mean()),
        "first_divergence": {str(b): v for b, v in first.items()},
        "mel_max_abs_from_reference_tokens": float(mel_err.max()), "wave_rms_from_reference_tokens": float(wave_rms.max()),
        "wave_rms_full_length_2_clips": full_rms, "wave_rms_last_32768_samples": tail_rms,
        "e2e_mel_max_abs_identical_clips": max([e for e, s in zip(e2e_mel, same) if s], default=None),
        "e2e_wave_rms_identical_clips": max([e for e, s in zip(e2e_rms, same) if s], default=None)})
    assert mel_err.max() < MEL_TOL and wave_rms.max() < WAVE_RMS_TOL
    assert full_rms < WAVE_RMS_TOL and tail_rms < WAVE_RMS_TOL
    for b in range(B):
        if same[b]:
            assert e2e_mel[b] < MEL_TOL and e2e_rms[b] < WAVE_RMS_TOL
        else:
            assert first[b]["near_tie"], "clip %d left the reference trajectory away from a near-tie: %s" % (b, first[b])
    assert sum(same) >= MIN_EXACT_CLIPS, "only %d of %d clips reproduce the reference's tokens" % (sum(same), B)


# ---- "same TEXT + seed": the chain entered at the caption STRINGS (BASELINE.json north_star) ------------------------------
# traj_T100_L19 is a reference run from caption strings: oracle/make_golden.py traj_full() tokenises 8 captions with the
# reference's Tokenize, embeds them with its fp16 CLIPTextEmbedding (the golden stores tokens and embedding) and runs the
# 100-step loop on that embedding.  The tests above enter at `cond_emb`; this one enters at the strings: this package's
# tokenizer -> the HIP CLIP tower (fp16 semantics) -> the same loop.  The tower is pinned to the reference to TEXT_COND_TOL on
# unit-norm rows (an fp16 pipeline is reproducible to about that across devices / library builds, the reference's own
# included); what that embedding difference does to the 212 000 decisions is MEASURED here and reported.  Measured on MI355X
# (round 4, profiles/r04f_new_tests.log): embedding max-abs 4.3e-4, 0 of 212 000 teacher-forced decisions differ, 8 of 8
# free-running clips end on the reference's tokens, mel 2.5e-5, waveform RMS 3.6e-7 -- the gates sit just above that.
TEXT_COND_TOL = 5e-4
TEXT_MAX_FLIPS = 4              # teacher-forced disagreements from the strings (measured: 0)
TEXT_MIN_EXACT_CLIPS = 7        # free-running clips (of 8) that must end on the reference's tokens (measured: 8)


def test_same_text_and_seed_from_caption_strings(model, voc, g):
    from text_to_sound_synthesis_amd.config import build_model, default_config
    with open(os.path.join(ROOT, "tests", "golden", "traj_T100_L19_captions.json")) as f:
        captions = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys_clip.json")) as f:
        clip_sd = synth.synth_state_dict(json.load(f))      # the weights oracle/ref_harness.py build_clip_text() gave the reference
    text = build_model(default_config(n_layer=1, with_clip=True))
    _, unexpected = text.load_state_dict(clip_sd, strict=False)
    assert not unexpected
    text = text.cuda().eval()
    # (clip.tokenize semantics on the closed-vocabulary merge table shipped with the package -- the part of CLIP's table the
    #  synthetic captions use; the 1.3 MB full table is not on the GPU box.  bench.py tokenises the same way.)
    from text_to_sound_synthesis_amd import tokenizer as tz
    ids = tz.tokenize(captions, context_length=77, add_start_and_end=True,
                      tokenizer=tz.SimpleTokenizer(bpe_path=tz.CLOSED_VOCAB_PATH))["token"]
    assert torch.equal(ids.long().cpu(), g["caption_tokens"].long()), "BPE ids differ from the reference's"
    cond = text.transformer.condition_emb(ids.cuda()).float()
    ref_cond = g["cond_emb"].float()
    cond_err = (cond.cpu() - ref_cond).abs().max().item()
    set_precision(model, "f16x2")
    dt = model.transformer
    trace = g["step_tokens"].long()
    B = trace.shape[1]
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips, gaps, cuts = 0, [], []
    for i in range(100):
        t = 99 - i
        x_t = torch.full((B, 265), 256, dtype=torch.long) if i == 0 else trace[i - 1]
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(),
                                 noise(t, (B, 257, 265)).cuda(), initial=(i == 0)).cpu()
        for b, p in (tok != trace[i]).nonzero().tolist():
            flips += 1
            gaps.append(float(g["gap"][i, b, p]))
            cuts.append(float(g["tmargin"][i, b, p]))
    agree = 1.0 - flips / (100.0 * B * 265)
    out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0, noise_fn=noise)
    tokens = out["content_token"].cpu()
    same = [bool(torch.equal(tokens[b], g["tokens"][b].long())) for b in range(B)]
    mel = model.decode_to_img(tokens.cuda(), (B, 256, 5, 53))
    wave = voc(mel[:, 0], scale=0.5, shift=0.5)
    n = g["wave_head"].shape[1]
    e2e_mel = [(mel[b, 0].cpu() - g["mel"][b]).abs().max().item() for b in range(B) if same[b]]
    e2e_rms = [(wave[b, 0, :n].cpu() - g["wave_head"][b]).pow(2).mean().sqrt().item() for b in range(B) if same[b]]
    report("f16x2", "from_caption_strings", {
        "bpe_ids_identical": True, "clip_cond_max_abs_vs_reference": cond_err,
        "teacher_forced_decisions": 100 * B * 265, "teacher_forced_flips": flips, "teacher_forced_agreement": agree,
        "largest_argmax_margin_of_a_flip": max(gaps, default=None), "largest_cut_margin_of_a_flip": max(cuts, default=None),
        "free_running_clips_with_identical_tokens": sum(same),
        "free_running_token_agreement": float((tokens == g["tokens"].long()).float().mean()),
        "e2e_mel_max_abs_identical_clips": max(e2e_mel, default=None), "e2e_wave_rms_identical_clips": max(e2e_rms, default=None)})
    assert cond_err < TEXT_COND_TOL
    assert flips <= TEXT_MAX_FLIPS, "%d flips in %d teacher-forced decisions from the caption strings" % (flips, 100 * B * 265)
    assert sum(same) >= TEXT_MIN_EXACT_CLIPS, "only %d of %d clips reproduce the reference's tokens from the strings" % (sum(same), B)
    for m_, r_ in zip(e2e_mel, e2e_rms):
        assert m_ < MEL_TOL and r_ < WAVE_RMS_TOL


# ---- round 5: the two configurations the goldens above did not pin at chain level -----------------------------------------
# (a) BASELINE configs[3] -- the 512-entry codebook (configs/caps_512.yaml:12,82: 513 classes, logits N = 512, a 512-key sort
#     in the top-r cut): tests/golden/traj_T100_L19_k512.npz = the reference's 100-step loop + decode + vocoder on 8 captions
#     (oracle/make_golden.py traj_k512(), same hooks and noise keys as traj_full()).
# (b) the benchmarked batch itself -- 64 DISTINCT captions (the B = 64 tests above replicate 8 captions 8 times):
#     tests/golden/traj_T100_L19_b64.npz = the reference's first 10 reverse steps on 64 captions (traj_b64()).
@pytest.fixture(scope="module")
def model_k512():
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=19, diffusion_step=100, n_embed=512))
    missing, unexpected = m.load_state_dict(dict(synth_sd("dalle_k512", 19)), strict=False)
    assert not unexpected
    m = m.cuda().eval()
    m.transformer.truncation_r = 0.85
    return m


def test_k512_chain_100_steps_vs_reference(model_k512, voc):
    g = golden("traj_T100_L19_k512")
    m = model_k512
    set_precision(m, "f16x2")
    dt = m.transformer
    cond = g["cond_emb"].float().cuda()
    trace = g["step_tokens"].long()                      # [100, 8, 265]
    B = trace.shape[1]
    assert int(trace.max()) <= 512 and int(g["tokens"].max()) < 512
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips = []
    for i in range(100):
        t = 99 - i
        x_t = torch.full((B, 265), 512, dtype=torch.long) if i == 0 else trace[i - 1]
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(),
                                 noise(t, (B, 513, 265)).cuda(), initial=(i == 0)).cpu()
        for b, p in (tok != trace[i]).nonzero().tolist():
            flips.append({"t": t, "clip": b, "pos": p, "gap": float(g["gap"][i, b, p]), "tmargin": float(g["tmargin"][i, b, p])})
    out = dt.sample(condition_token=None, condition_mask=None, condition_embed=cond, filter_ratio=0, noise_fn=noise)
    tokens = out["content_token"].cpu()
    same = [bool(torch.equal(tokens[b], g["tokens"][b].long())) for b in range(B)]
    mel = m.decode_to_img(g["tokens"].long().cuda(), (B, 256, 5, 53))
    mel_err = float((mel[:, 0].cpu() - g["mel"]).abs().max())
    wave = voc(mel[:, 0], scale=0.5, shift=0.5)
    n = g["wave_head"].shape[1]
    wave_rms = float((wave[:, 0, :n].cpu() - g["wave_head"]).pow(2).mean(1).sqrt().max())
    full_rms, tail_rms = full_wave_rms(g, wave)
    report("f16x2", "k512_chain", {"decisions": 100 * B * 265, "teacher_forced_flips": len(flips),
                                   "free_running_clips_with_identical_tokens": sum(same), "clips": B,
                                   "mel_max_abs_from_reference_tokens": mel_err, "wave_rms_from_reference_tokens": wave_rms,
                                   "wave_rms_full_length_2_clips": full_rms, "wave_rms_last_32768_samples": tail_rms,
                                   "detail": flips[:16]})
    assert full_rms < WAVE_RMS_TOL and tail_rms < WAVE_RMS_TOL
    unexplained = [f for f in flips if not (f["tmargin"] < CUT_TIE or f["gap"] < GAP_TIE)]
    assert not unexplained, "K = 512: token disagreements away from any near-tie: %s" % unexplained[:4]
    assert len(flips) <= MAX_FLIPS["f16x2"]
    assert sum(same) >= MIN_EXACT_CLIPS
    assert mel_err < MEL_TOL and wave_rms < WAVE_RMS_TOL


def test_batch64_distinct_captions_first_10_steps_vs_reference(model):
    """The benchmarked batch (64 distinct captions, padded-row mode, per-sample GEMM program) teacher-forced on the
    reference's own first 10 reverse steps: 169 600 decisions against the reference itself, no replicas."""
    g = golden("traj_T100_L19_b64")
    set_precision(model, "f16x2")
    dt = model.transformer
    assert dt.transformer.row_padding
    cond = g["cond_emb"].float().cuda()
    trace = g["step_tokens"].long()                      # [10, 64, 265]
    S, B = trace.shape[0], trace.shape[1]
    assert B == 64 and len({tuple(r.tolist()) for r in g["caption_tokens"]}) == 64      # 64 DISTINCT captions
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips = []
    for i in range(S):
        t = 99 - i
        x_t = torch.full((B, 265), 256, dtype=torch.long) if i == 0 else trace[i - 1]
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(),
                                 noise(t, (B, 257, 265)).cuda(), initial=(i == 0)).cpu()
        for b, p in (tok != trace[i]).nonzero().tolist():
            flips.append({"t": t, "clip": b, "pos": p, "gap": float(g["gap"][i, b, p]), "tmargin": float(g["tmargin"][i, b, p])})
    report("f16x2", "batch64_distinct_captions", {"decisions": S * B * 265, "flips": len(flips), "detail": flips[:16]})
    unexplained = [f for f in flips if not (f["tmargin"] < CUT_TIE or f["gap"] < GAP_TIE)]
    assert not unexplained, "token disagreements away from any near-tie: %s" % unexplained[:4]
    assert len(flips) <= MAX_FLIPS["f16x2"]


# ---- round 6: the chain OFF the N(0, 0.02) manifold -------------------------------------------------------------------------
# Every chain golden above runs initialiser-like weights: near-uniform logits (the worst case for argmax flips), but activations
# O(1) and a top-r cut that keeps ~200 classes.  tests/golden/traj_T100_L19_trainedlike.npz (oracle/make_golden.py traj_trained())
# is the reference's 100-step loop on the TRAINED-LIKE denoiser of synth.py (GELU2 outputs in the 1e4s -- fp16 tops out at 65504
# --, four residual channels ~100x hot, LayerNorm gains over two decades, Student-t weights): peaky posteriors whose cut falls
# after a handful of ranks (19-32 distinct codes per final clip against ~130 above) and a dynamic range where an fp16 plane can
# saturate and a lo plane go subnormal.  On these weights the reference's OWN fp32 logits are 1.6e-3 away from its float64 ones
# (transformer_L19_trainedlike.npz: fp32_vs_fp64), so a decision whose margin is inside what an error of that size can move may
# fall either way on any fp32 implementation -- the reference on another BLAS included.  The bands below are that distance
# (x 20 on the Gumbel-argmax margin, as tests/test_hip_trained_like.py uses for the single step; the cut margin is a
# probability mass, where a logit error e moves the mass ranked before a class by up to ~e).
TRAINED_GAP_TIE = 20 * 1.6e-3
TRAINED_CUT_TIE = 2 * 1.6e-3
TRAINED_MAX_FLIPS = 400         # of 212 000 teacher-forced decisions, every one inside the bands (provisional: set from the first GPU run)


@pytest.fixture(scope="module")
def model_trained():
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=19, diffusion_step=100))
    missing, unexpected = m.load_state_dict(dict(synth_sd("dalle", 19, profile="trained")), strict=False)
    assert not unexpected
    m = m.cuda().eval()
    m.transformer.truncation_r = 0.85
    return m


@pytest.mark.parametrize("mode", ["fp32", "f16x2"])
def test_trained_like_chain_100_steps_vs_reference(model_trained, voc, mode):
    g = golden("traj_T100_L19_trainedlike")
    ref_err = float(golden("transformer_L19_trainedlike")["fp32_vs_fp64"])
    assert ref_err < 2e-3                                  # what the bands above are derived from
    m = model_trained
    set_precision(m, mode)
    dt = m.transformer
    cond = g["cond_emb"].float().cuda()
    trace = g["step_tokens"].long()
    B = trace.shape[1]
    kv = dt.transformer.condition_kv(cond, dt._schedule_table())
    flips = []
    for i in range(100):
        t = 99 - i
        x_t = torch.full((B, 265), 256, dtype=torch.long) if i == 0 else trace[i - 1]
        tok = dt.p_sample_tokens(x_t.cuda(), kv, torch.full((B,), t, dtype=torch.long).cuda(),
                                 noise(t, (B, 257, 265)).cuda(), initial=(i == 0)).cpu()
        for b, p in (tok != trace[i]).nonzero().tolist():
            flips.append({"t": t, "clip": b, "pos": p, "ours": int(tok[b, p]), "ref": int(trace[i, b, p]),
                          "gap": float(g["gap"][i, b, p]), "tmargin": float(g["tmargin"][i, b, p])})
    # the free-running chain: how many clips stay on the reference's trajectory, and where the others leave it
    x = torch.full((B, 265), 256, dtype=torch.long).cuda()
    first = {}
    for i in range(100):
        t = 99 - i
        x = dt.p_sample_tokens(x, kv, torch.full((B,), t, dtype=torch.long).cuda(), noise(t, (B, 257, 265)).cuda(), initial=(i == 0))
        xc = x.cpu()
        for b in range(B):
            if b not in first and not torch.equal(xc[b], trace[i, b]):
                pos = (xc[b] != trace[i, b]).nonzero().flatten().tolist()
                first[b] = {"t": t, "pos": pos[:8],
                            "near_tie": all(float(g["tmargin"][i, b, p]) < TRAINED_CUT_TIE or float(g["gap"][i, b, p]) < TRAINED_GAP_TIE
                                            for p in pos)}
    tokens = x.cpu()
    same = [bool(torch.equal(tokens[b], g["tokens"][b].long())) for b in range(B)]
    mel = m.decode_to_img(g["tokens"].long().cuda(), (B, 256, 5, 53))
    mel_err = float((mel[:, 0].cpu() - g["mel"]).abs().max())
    wave = voc(mel[:, 0], scale=0.5, shift=0.5)
    full_rms, tail_rms = full_wave_rms(g, wave)
    strict = [f for f in flips if not (f["tmargin"] < CUT_TIE or f["gap"] < GAP_TIE)]        # outside the init-profile bands
    unexplained = [f for f in flips if not (f["tmargin"] < TRAINED_CUT_TIE or f["gap"] < TRAINED_GAP_TIE)]
    report(mode, "trained_like_chain", {
        "decisions": 100 * B * 265, "teacher_forced_flips": len(flips), "flips_outside_init_profile_bands": len(strict),
        "flips_outside_trained_bands": len(unexplained),
        "largest_argmax_margin_of_a_flip": max((f["gap"] for f in flips), default=None),
        "free_running_clips_with_identical_tokens": sum(same), "clips": B,
        "free_running_token_agreement": float((tokens == g["tokens"].long()).float().mean()),
        "first_divergence": {str(b): v for b, v in first.items()},
        "mel_max_abs_from_reference_tokens": mel_err, "wave_rms_full_length_2_clips": full_rms,
        "wave_rms_last_32768_samples": tail_rms, "reference_fp32_vs_float64_logits": ref_err, "detail": flips[:16]})
    assert not unexplained, "trained-like chain: disagreements away from any near-tie: %s" % unexplained[:4]
    assert len(flips) <= TRAINED_MAX_FLIPS
    for b in range(B):
        if not same[b]:
            assert first[b]["near_tie"], "clip %d left the reference trajectory away from a near-tie: %s" % (b, first[b])
    assert mel_err < MEL_TOL and full_rms < WAVE_RMS_TOL and tail_rms < WAVE_RMS_TOL
