"""Training-iteration timer for BASELINE.json configs[4] / SURVEY.md section 8d row 5 -- the measurement behind bench.py's
`train` object and tools/bench_train.py's command line.

One iteration = what the reference's trainer does per batch (engine/solver_spec.py:308-331 over
models/dalle_spec.py:93-133,389-400 and transformers/diffusion_transformer.py:408-476,539-577), FROM THE REFERENCE'S
BATCH:

    {'image': mel ~ U(-1, 1) f32[B,1,80,848], 'text': B synthetic captions}
      -> BPE ids (host) -> CLIP text tower -> f32[B,77,512]          |  modeling.train.training_inputs
      -> VQ encoder + nearest code + ColumnMajor -> i64[B,265]       |  (eager, on the replay's stream)
      -> sample_time -> q_sample -> 19-layer forward keeping activations -> loss -> hand-written backward
      -> (bucketed RCCL all-reduce) -> global-norm clip -> AdamW     |  one hipGraph (two with N ranks)
      -> LR schedule -> EMA                                          |  host-driven, the reference's order

A new mel (drawn on the device) and new captions every iteration.  Reported for a run of `steps` iterations after `warmup`
(the warm-up holds the calibration backward and the graph capture):

    it_per_s_sustained   steps / wall time of the whole run, re-captures included -- `value`
    it_per_s_replay      1 / median device time between the starts of consecutive iterations (HIP events on the stream;
                         an interval that holds a re-capture is an outlier the median drops)
    recaptures           how often the captured iteration was rebuilt in the run, and why (`recapture_reasons`)
    monitor_log2         the saturation monitor's readings: log2 max |scaled dY| per 16 iterations
"""
import time

import torch
import torch.distributed as dist


def run(batch=20, steps=200, warmup=5, n_layer=19, codes=256, precision="f16x2", ema_device="cuda", attention="fused",
        graph=True, world=1, rank=0, dev=None, monitor_hi=None, from_batch=True, calib_target=None, prefetch=False,
        profile="init"):
    """Time `steps` training iterations (after `warmup`) and return the result dict (module docstring).
    from_batch=False: the round-5 form -- pre-made tokens and a stand-in caption embedding, no mel / caption prologue
    (kept as the A/B leg that prices the prologue).  (Per-kernel rates: tools/train_profile.sh.)"""
    from . import shard, synth, tokenizer
    from .config import build_model, default_config
    from .modeling.solver import EMA, GradClipWindow, GraphSolver, PlateauWarmupLR, Solver
    from .modeling.train import TrainStep

    # (the synthetic captions tokenise on the closed-vocabulary merge table shipped with the package, as in bench.py's sampling
    #  loop: CLIP's 1.3 MB full table is not on the GPU box)
    m = build_model(default_config(n_layer=n_layer, diffusion_step=100, n_embed=codes, with_clip=from_batch,
                                   bpe_path=tokenizer.CLOSED_VOCAB_PATH))
    synth.synth_init_(m, seed=0)
    if profile != "init":            # the denoiser off the N(0, 0.02) manifold (synth.py profile="trained": heavy-tailed gradients)
        synth.synth_init_(m, seed=0, skip=("content_codec.", "transformer.condition_emb."), profile=profile)
    m = m.to(dev).eval()
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]   # configs/caps.yaml
    B, K1, L = batch, codes + 1, 265
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    if not from_batch:
        x0 = synth.synth_tokens(B, L, codes, mask_frac=0.0, key="bt.x0.%d" % rank).to(dev)
        cond = synth.synth_cond_emb(B, key="bt.c.%d" % rank).to(dev)

    times = {"grads": 0.0, "allreduce": 0.0, "update": 0.0}
    timing = [False]

    def timed_allreduce(grads):
        if timing[0] and not graph:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        shard.allreduce_gradients(grads)
        if timing[0] and not graph:
            torch.cuda.synchronize()
            times["allreduce"] += time.perf_counter() - t0

    class Timed(TrainStep):
        def loss_and_grads(self, *a, **k):
            if timing[0]:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = super().loss_and_grads(*a, **k)
            if timing[0]:
                torch.cuda.synchronize()
                times["grads"] += time.perf_counter() - t0
            return out

    sched = PlateauWarmupLR(3.0e-6, factor=0.5, patience=25000, min_lr=1.0e-6, threshold=1.0e-1, warmup_lr=4.5e-4, warmup=1000)
    ema = EMA(dt, decay=0.99, update_interval=25, device=ema_device)
    use_graph = bool(graph)
    step = (TrainStep if use_graph else Timed)(dt, precision=precision, attention=attention)
    if monitor_hi is not None:           # experiment: the upper bound (log2) of the saturation monitor's window
        step.monitor_window = (step.monitor_window[0], monitor_hi)
    if calib_target is not None:         # experiment: where the calibration puts the largest |scaled dY| (log2)
        step.calib_log2 = calib_target
    common = dict(lr=3.0e-6, betas=(0.9, 0.96), weight_decay=4.5e-2, scheduler=sched, clip_grad_norm=GradClipWindow(0, 5000, 0.5),
                  ema=ema, model=m, generator=gen)
    if use_graph:
        # one GPU: the whole iteration is one hipGraph.  Data parallel: two graphs per rank (gradients | clip + AdamW) with the
        # bucketed all-reduce over RCCL enqueued between the replays (tests/test_hip_rccl.py runs exactly this at world 1)
        solver = GraphSolver(step, reduce=timed_allreduce if world > 1 else None, **common)
    else:
        solver = Solver(step, allreduce=timed_allreduce if world > 1 else None, **common)

    it_no = [0]
    ahead = []

    data_stream = torch.cuda.Stream(dev) if (prefetch and from_batch) else None      # where the synthetic mel is drawn
    gen_data = torch.Generator(device=dev).manual_seed(4321 + rank)

    def make_batch():
        it_no[0] += 1
        if data_stream is not None:
            with torch.cuda.stream(data_stream):
                mel = torch.rand((B, 1, 80, 848), device=dev, generator=gen_data) * 2.0 - 1.0
        else:
            mel = torch.rand((B, 1, 80, 848), device=dev, generator=gen) * 2.0 - 1.0
        return {"image": mel, "text": synth.synth_captions(B, seed=100003 * rank + it_no[0])}

    def one():
        if from_batch and prefetch:
            # software pipeline: the NEXT batch's prologue is enqueued on a side stream right after this iteration's replay
            if ahead:
                cur = ahead.pop()
            else:
                cur = make_batch()
                torch.cuda.current_stream(dev).wait_stream(data_stream)
            out_ = solver.step(cur)
            nxt = make_batch()
            solver.prefetch(nxt, ready=data_stream)
            ahead.append(nxt)
            return out_
        if from_batch:
            return solver.step(make_batch())
        t, pt = dt.sample_time(B, dev, "importance", generator=gen)
        u = torch.rand((B, K1, L), device=dev, generator=gen)
        return solver.step(x0, cond, t, pt, u)

    for _ in range(warmup):
        out = one()
    g = getattr(solver, "iteration_graph", None)
    rec0 = getattr(g, "recaptures", 0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing[0] = True
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for i in range(steps):
        marks[i].record()
        out = one()
    marks[steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = el.item()
    timing[0] = False
    times["update"] = el - times["grads"] - times["allreduce"]
    gaps = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median_ms = gaps[len(gaps) // 2]
    g = getattr(solver, "iteration_graph", None)
    return {
        "metric": "training iterations/s, sustained (mel + captions -> BPE, CLIP, VQ encode -> loss + backward + clip + AdamW + EMA)"
                  if from_batch else "training iterations/s (denoiser step from tokens: loss + backward + clip + AdamW + EMA)",
        "value": steps / el, "it_per_s_sustained": steps / el, "it_per_s_replay": 1e3 / median_ms,
        "unit": "it/s", "samples_per_s": steps * B * world / el, "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * el / steps, "ms_per_replay_median": median_ms,
        "ms_per_iteration_max": gaps[-1],
        "dtype": "f32 via 2-way fp16 split (linear layers fwd + dX + dW), fp32 elsewhere" if precision == "f16x2" else "f32",
        "data": "synthetic", "loss": float(out["loss"]), "grad_norm": float(out["grad_norm"]),
        "ms": {k: 1e3 * v / steps for k, v in times.items()},
        "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        "graph": use_graph, "attention": attention, "prefetch": bool(prefetch and from_batch), "weights": profile,
        "loss_scale_exp": solver.train_step.loss_scale_exp,
        # the saturation monitor over the run: log2 of max |scaled dY| at each check, and how often the iteration was re-captured
        "monitor_log2": list(step.monitor_log), "recaptures": getattr(g, "recaptures", 0) - rec0,
        "recapture_reasons": list(getattr(g, "recapture_reasons", []))[-8:],
        "config": {"workload": ("BASELINE configs[4]: training iteration from the reference's batch -- mel U(-1,1) f32[%d,1,80,848] + %d "
                                "captions -> BPE + CLIP + VQ encode -> loss + backward + clip + AdamW + EMA, %d layers, K=%d"
                                % (B, B, n_layer, codes)) if from_batch else
                               "training step from pre-made tokens (no mel / caption prologue), B=%d per GPU, %d layers, K=%d"
                               % (B, n_layer, codes),
                   "parallelism": "dp%d" % world}}
