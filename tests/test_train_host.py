"""Host composition of the training step (modeling/train.py, scope row 8f-3) on the CPU: the C-ABI entries it calls
are replaced by tests/hip_abi_emulation.py (torch restatements of the header's contracts), so what is tested here is
everything that is NOT a kernel -- fused Q|K|V / K|V projections and their gradient slices, operand layouts and
paddings handed to the GEMMs, the split-K dW launches, the loss scale of the "f16x2" backend (calibration, carrying it
through the backward, taking it out again), AdaLN table gradients, the gradient dict's names -- against autograd
through the oracle, which the CPU suite pins to the reference's loss.backward() (test_oracle_golden.py).  The kernels
themselves are tested against the same oracle on the GPU (tests/test_hip_train_kernels.py)."""
import pytest
import torch

import diffsound_oracle as O
import hip_abi_emulation
from conftest import golden, synth_sd
from text_to_sound_synthesis_amd import synth


def _model():
    from text_to_sound_synthesis_amd.config import build_model, default_config
    m = build_model(default_config(n_layer=2, diffusion_step=100))
    sd_cpu = dict(synth_sd("dalle", 2))
    m.load_state_dict({**sd_cpu, **synth_sd("encoder")}, strict=False)
    dt = m.eval().transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]
    return dt, sd_cpu


def _batch():
    return (synth.synth_tokens(3, mask_frac=0.0, key="tl.x0"), synth.synth_cond_emb(3, key="tl.c"),
            torch.tensor([57, 0, 93]), torch.ones(3) / 100, synth.synth_uniform((3, 257, 265), key="tl.u"))


@pytest.fixture(scope="module")
def oracle_grads():
    sd_cpu = dict(synth_sd("dalle", 2))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    x0, cond, t, pt, u = _batch()
    with torch.enable_grad():
        _, _, loss, lt2 = O.train_loss(sd, x0, cond, t, pt, u)
        loss.backward()
    return loss.item(), {k[len("transformer."):]: v.grad for k, v in sd.items()
                         if k.startswith("transformer.transformer.") and v.is_floating_point() and v.grad is not None}, lt2


@pytest.mark.parametrize("precision,attention", [("fp32", "fused"), ("f16x2", "fused"), ("f16x2", "composed")])
def test_training_step_host_composition(monkeypatch, oracle_grads, precision, attention):
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    hip_abi_emulation.install(monkeypatch)
    loss_ref, want, lt2 = oracle_grads
    dt, _ = _model()
    step = TrainStep(dt, precision=precision, attention=attention)
    batch = _batch()
    loss, grads = step.loss_and_grads(*batch)
    assert abs(loss.item() - loss_ref) < 2e-5 * loss_ref
    assert abs(loss.item() - float(golden("train_loss_L2")["loss"])) < 2e-4 * loss_ref      # the reference's own value
    if precision == "f16x2":
        # calibration: the largest |dY| entering a GEMM sits at 2^12 .. 2^13 after scaling
        k = step.loss_scale_exp
        assert 2.0 ** 12 <= step.calibrated_amax * 2.0 ** k < 2.0 ** 13 and k > 0
    worst = []
    missing = [n for n, g in want.items() if n not in grads and g.abs().max() > 0]
    assert not missing, missing
    for name, g in want.items():
        if name not in grads:
            continue
        got = grads[name].double()
        assert got.shape == g.shape, name
        if g.abs().max().item() < 1e-7:             # analytically zero gradients (key biases): rounding noise on both sides
            assert got.abs().max().item() < 1e-6, name
            continue
        worst.append(((got - g.double()).abs().max().item() / g.abs().max().item(), name))
    worst.sort(reverse=True)
    print("precision %s: worst per-tensor gradient error %.2e (%s)" % (precision, worst[0][0], worst[0][1]))
    assert len(worst) >= 50 and worst[0][0] < 2e-4, worst[:5]
    # the importance-sampling statistics were updated exactly once (the calibration pass must not touch them)
    cnt = torch.zeros(100)
    cnt[batch[2]] = 1.0
    assert torch.equal(dt.Lt_count, cnt)
    assert torch.allclose(dt.Lt_history[batch[2]], 0.1 * lt2.detach(), rtol=1e-4)
    # one AdamW update through the device-scalar entry == the scalar-argument entry
    import math
    names = ["transformer.to_logits.1.weight", "transformer.blocks.0.attn1.key.weight", "transformer.blocks.1.ln2.weight"]
    params = dict(dt.named_parameters())
    before = {n: params[n].detach().clone() for n in names}
    step.adamw_step({n: grads[n] for n in names}, {}, 1, lr=1e-3)
    after_a = {n: params[n].detach().clone() for n in names}
    for n in names:
        params[n].data.copy_(before[n])
    hyper = torch.tensor([1e-3, 1 - 0.9, math.sqrt(1 - 0.96), 1.0])
    step.adamw_step({n: grads[n] for n in names}, {}, 0, 0.0, hyper=hyper)
    for n in names:
        assert not torch.equal(after_a[n], before[n])
        assert torch.allclose(params[n].detach(), after_a[n], rtol=0, atol=1e-7), n


def test_gradients_are_handed_over_during_the_backward(monkeypatch):
    """loss_and_grads(on_grads=...): the hook of the overlapped data-parallel reduction (shard.GradientReducer.ready) is
    called after the logits layer and after every block, last block first, with weight-matrix gradients that are FINAL at
    that moment (a copy taken inside the hook equals the returned gradient), each name once; what is never handed over is
    the small rest (biases, norm gains, embedding tables) that the step un-scales at its end."""
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    hip_abi_emulation.install(monkeypatch)
    dt, _ = _model()
    step = TrainStep(dt, precision="f16x2")
    calls, snap = [], {}

    def hook(named, streams):
        calls.append(sorted(named))
        for n, t in named.items():
            assert n not in snap
            snap[n] = t.clone()
    loss, grads = step.loss_and_grads(*_batch(), on_grads=hook)
    assert calls[0] == ["transformer.to_logits.1.weight"] and len(calls) == 1 + 2
    assert all(n.startswith("transformer.blocks.1.") for n in calls[1]) and all(n.startswith("transformer.blocks.0.") for n in calls[2])
    assert len(calls[1]) == 16
    for n, t in snap.items():
        assert torch.equal(t, grads[n]), n
    handed = sum(t.numel() for t in snap.values())
    total = sum(t.numel() for t in grads.values())
    assert handed / total > 0.97
    rest = [n for n in grads if n not in snap]
    assert all(n.endswith((".bias", "ln2.weight", "to_logits.0.weight", "emb.weight")) for n in rest), rest[:5]


def test_split_k_plan_and_padding():
    from text_to_sound_synthesis_amd.modeling.train import _SplitGemm, _ceil
    # (N, K) of the denoiser's linears -> K-ranges; the padded contraction length divides into 32-wide k-tiles per range
    for (N, K), want in {(1024, 1024): 8, (3072, 1024): 4, (4096, 1024): 4, (1024, 4096): 4, (2048, 512): 8, (256, 1024): 8}.items():
        S = _SplitGemm.split_k(N, K)
        assert S == want, (N, K, S)
        for M in (231, 795, 5300, 1540):
            Mp = _ceil(M, 32 * S)
            assert Mp >= M and (Mp // S) % 32 == 0 and Mp - M < 32 * S


def test_loss_scale_guard():
    """observe_grad_norm: the calibrated loss scale is kept inside a (1/64, 4) window of the gradient norm at calibration."""
    from text_to_sound_synthesis_amd.modeling.train import TrainStep
    dt, _ = _model()
    step = TrainStep(dt, precision="f16x2")
    step.loss_scale_exp = 17
    assert not step.observe_grad_norm(3.0)             # first observation = the reference
    assert not step.observe_grad_norm(11.0) and not step.observe_grad_norm(0.06) and step.loss_scale_exp == 17
    assert step.observe_grad_norm(12.5) and step.loss_scale_exp is None       # > 4x: re-calibrate on the next step
    step.loss_scale_exp = 15
    assert not step.observe_grad_norm(12.5)            # new reference
    assert step.observe_grad_norm(0.1) and step.loss_scale_exp is None        # < 1/64
    fp = TrainStep(dt, precision="fp32")
    assert not fp.observe_grad_norm(1.0) and not fp.observe_grad_norm(1e9) and fp.loss_scale_exp == 0
