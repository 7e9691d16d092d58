"""Command line of the training-iteration timer (BASELINE.json configs[4] / SURVEY.md section 8d row 5).  The measurement itself
lives in the package -- text_to_sound_synthesis_amd/train_bench.py: run() -- and is what bench.py's `train` object reports.

  python tools/bench_train.py --graph --steps 200                  (the contracted iteration: mel + captions in)
  python tools/bench_train.py --graph --steps 200 --from-tokens    (round-5 form: no mel / caption prologue -- prices the prologue)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --graph ...

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL; before any HIP call
    import torch
    import torch.distributed as dist
    from text_to_sound_synthesis_amd import train_bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=20, help="samples per GPU (configs/caps.yaml:136)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-layer", type=int, default=19)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--precision", default="f16x2", choices=("f16x2", "fp32"),
                    help="linear-layer GEMMs (forward, dX, dW): 3-pass fp16 split or exact-fp32 MFMA")
    ap.add_argument("--ema-device", default="cuda", help="the reference keeps the EMA on the CPU (configs/caps.yaml:101)")
    ap.add_argument("--attention", default="fused", choices=("fused", "composed"),
                    help="fused: ds_attention + ds_attention_bwd (recompute); composed: grouped GEMMs with stored probabilities")
    ap.add_argument("--monitor-hi", type=int, default=None, help="experiment: log2 upper bound of the saturation monitor's window")
    ap.add_argument("--calib-log2", type=int, default=None, help="experiment: log2 of where a calibration puts the largest |dY|")
    ap.add_argument("--from-tokens", action="store_true", help="pre-made tokens + stand-in caption embedding (no prologue)")
    ap.add_argument("--weights", default="init", choices=("init", "trained"), help="synth.py weight profile of the denoiser")
    ap.add_argument("--prefetch", action="store_true", help="the next batch's BPE / CLIP / VQ-encode prologue on a side stream")
    ap.add_argument("--graph", action="store_true", help="gradients -> clip -> AdamW captured in one hipGraph (one GPU)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    out = train_bench.run(args.batch, args.steps, args.warmup, args.n_layer, args.codes, args.precision, args.ema_device,
                          args.attention, args.graph, world, rank, dev, args.monitor_hi, from_batch=not args.from_tokens,
                          calib_target=args.calib_log2, prefetch=args.prefetch, profile=args.weights)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
