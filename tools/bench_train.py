"""Training-iteration timer for BASELINE.json configs[4] / SURVEY.md section 8d row 5 (NOT the driver's bench: that is
bench.py at the repo root).  One iteration = DALLE-side training step of the denoiser on an AudioCaps-shaped synthetic
batch: q_sample -> 19-layer forward keeping activations -> loss -> hand-written backward -> (bucketed RCCL all-reduce)
-> global-norm clip -> AdamW -> LR schedule -> EMA, i.e. modeling/solver.py: Solver.step on modeling/train.py: TrainStep.

  python tools/bench_train.py --batch 20 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py ...

  python tools/bench_train.py --precision f16x2 --graph        (the iteration replayed as one hipGraph; N ranks: two graphs with
                                                                the RCCL all-reduce between them)

Prints one JSON line on rank 0: iterations/s, samples/s (whole job), ms per phase (loss+gradients / all-reduce / update;
with --graph everything is one launch and only the total is meaningful).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=20, steps=5, warmup=2, n_layer=19, codes=256, precision="f16x2", ema_device="cuda", attention="fused",
        graph=True, world=1, rank=0, dev=None, monitor_hi=None):
    """Time `steps` training iterations (after `warmup`) and return the result dict (see the module docstring).  (Per-kernel
    rates: tools/train_profile.sh -- rocprofv3 --stats over this script.)"""
    from text_to_sound_synthesis_amd import shard, synth
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.solver import EMA, GradClipWindow, PlateauWarmupLR, Solver
    from text_to_sound_synthesis_amd.modeling.train import TrainStep

    m = build_model(default_config(n_layer=n_layer, diffusion_step=100, n_embed=codes))
    synth.synth_init_(m, seed=0)
    m = m.to(dev).eval()
    dt = m.transformer
    dt.auxiliary_loss_weight, dt.adaptive_auxiliary_loss, dt.mask_weight = 5.0e-4, True, [1, 1]   # configs/caps.yaml
    B, K1, L = batch, codes + 1, 265
    x0 = synth.synth_tokens(B, L, codes, mask_frac=0.0, key="bt.x0.%d" % rank).to(dev)
    cond = synth.synth_cond_emb(B, key="bt.c.%d" % rank).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    times = {"grads": 0.0, "allreduce": 0.0, "update": 0.0}
    timing = [False]

    def timed_allreduce(grads):
        if timing[0]:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        shard.allreduce_gradients(grads)
        if timing[0]:
            torch.cuda.synchronize()
            times["allreduce"] += time.perf_counter() - t0

    class Timed(TrainStep):
        def loss_and_grads(self, *a):
            if timing[0]:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = super().loss_and_grads(*a)
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
    if use_graph:
        # one GPU: the whole iteration is one hipGraph.  Data parallel: two graphs per rank (gradients | clip + AdamW) with the
        # bucketed all-reduce over RCCL enqueued between the replays (tests/test_hip_rccl.py runs exactly this at world 1)
        from text_to_sound_synthesis_amd.modeling.solver import GraphSolver
        solver = GraphSolver(step, lr=3.0e-6, betas=(0.9, 0.96), weight_decay=4.5e-2,
                             scheduler=sched, clip_grad_norm=GradClipWindow(0, 5000, 0.5), ema=ema,
                             reduce=timed_allreduce if world > 1 else None)
    else:
        solver = Solver(step, lr=3.0e-6, betas=(0.9, 0.96), weight_decay=4.5e-2,
                        scheduler=sched, clip_grad_norm=GradClipWindow(0, 5000, 0.5), ema=ema,
                        allreduce=timed_allreduce if world > 1 else None)

    def one():
        t, pt = dt.sample_time(B, dev, "importance")
        u = torch.rand((B, K1, L), device=dev, generator=gen)
        return solver.step(x0, cond, t, pt, u)

    for _ in range(warmup):
        out = one()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = el.item()
    timing[0] = False
    times["update"] = el - times["grads"] - times["allreduce"]
    return {
        "metric": "training iterations/s (denoiser step: loss + backward + clip + AdamW + EMA)", "value": steps / el,
        "unit": "it/s", "samples_per_s": steps * B * world / el, "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * el / steps,
        "dtype": "f32 via 2-way fp16 split (linear layers fwd + dX + dW), fp32 elsewhere" if precision == "f16x2" else "f32",
        "data": "synthetic", "loss": float(out["loss"]), "grad_norm": float(out["grad_norm"]),
        "ms": {k: 1e3 * v / steps for k, v in times.items()},
        "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        "graph": use_graph, "attention": attention,
        "loss_scale_exp": solver.train_step.loss_scale_exp,
        # the saturation monitor over the run: log2 of max |scaled dY| at each check, and how often the iteration was re-captured
        "monitor_log2": list(step.monitor_log), "recaptures": getattr(getattr(solver, "iteration_graph", None), "recaptures", 0),
        "config": {"workload": "BASELINE configs[4]: training step, B=%d per GPU, %d layers, K=%d" % (B, n_layer, codes),
                   "parallelism": "dp%d" % world}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=20, help="samples per GPU (configs/caps.yaml:136)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-layer", type=int, default=19)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--precision", default="fp32", choices=("f16x2", "fp32"),
                    help="linear-layer GEMMs (forward, dX, dW): 3-pass fp16 split or exact-fp32 MFMA")
    ap.add_argument("--ema-device", default="cuda", help="the reference keeps the EMA on the CPU (configs/caps.yaml:101)")
    ap.add_argument("--attention", default="fused", choices=("fused", "composed"),
                    help="fused: ds_attention + ds_attention_bwd (recompute); composed: grouped GEMMs with stored probabilities")
    ap.add_argument("--monitor-hi", type=int, default=None, help="experiment: log2 upper bound of the saturation monitor's window")
    ap.add_argument("--graph", action="store_true", help="gradients -> clip -> AdamW captured in one hipGraph (one GPU)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    out = run(args.batch, args.steps, args.warmup, args.n_layer, args.codes, args.precision, args.ema_device, args.attention,
              args.graph, world, rank, dev, args.monitor_hi)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
