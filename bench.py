"""Benchmark of the Diffsound generation path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Either the driver starts the ranks (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: WORLD_SIZE is in the environment), or
-- WORLD_SIZE unset -- bench.py starts them ITSELF the same way (launch_ranks below; the reference spawns its own workers
too, Diffsound/sound_synthesis/distributed/launch.py:26-57).  Asking for more ranks than there are devices, or a
WORLD_SIZE that contradicts --gpus, is an error, never a silent one-GPU run.

One "step" = one pass of the hot path over one batch: B captions per GPU go through the 100-step
p_sample loop (19-layer denoiser + fused sampler tail), SpecVQGAN decode and the MelGAN vocoder,
ending with f32[B, 1, 217088] waveforms in HBM.  Workload = BASELINE.json configs[2] (full pipeline,
batch 64 per GPU, K=256 codebook): synthetic caption strings -> BPE tokenizer (host, closed-vocabulary
merge table) -> CLIP text tower -> 100-step diffusion -> SpecVQGAN decode -> MelGAN.  Weights are
seeded random-init tensors of the reference's exact shapes (no checkpoints exist offline).
Multi-GPU: captions shard across ranks (weak scaling, fixed B per GPU); rank 0 scatters the caption
token ids and gathers the waveforms over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).  `roofline` is measured live:
HIP events around every denoiser GEMM launch in a separate profiled batch (ds_profile_*), algorithmic
flops 2*M*N*K per launch.  `cpu_baseline` times the reference itself (when /root/reference exists) or the CPU oracle (a
restatement of the reference path) on a bounded sample on this box's host cores, thread count swept.

Everything after the timed region is a side leg (run_side_legs): a leg that raises is filed as {"error": ...} under its
name, and a leg that hangs -- N ranks, one of them gone inside a collective -- is cut off by HeadlineGuard after
--side-leg-limit seconds (N > 1 only): the line is printed with what it has ("incomplete": reason) and every rank exits 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/fp16 dense MFMA peak (~2.5 PF)
GFLOP_PER_SAMPLE_STEP = 155.02  # SURVEY.md section 8d, hoisted formulation


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="captions per GPU per step")
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--n-layer", type=int, default=19)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--precision", default=os.environ.get("DIFFSOUND_GEMM", "f16x2"), choices=("fp32", "f16x2"),
                    help="denoiser GEMM arithmetic: f16x2 (default) = fp32-class split on the fp16 matrix cores, 2 fp16 planes per "
                         "operand, 3 MFMA passes; fp32 = the exact-fp32 MFMA (strict mode)")
    ap.add_argument("--rng", default="philox", choices=("philox", "torch"),
                    help="philox (default): Gumbel noise drawn in the sampler kernel, keyed by the global caption index -- "
                         "the same NOISE for a caption at every world size / batch (identical tokens as long as the same GEMM "
                         "program serves the local batch: logits differ by ~1e-7 relative between programs, so a near-tie "
                         "can flip); torch: torch.rand per step, the reference's own draw")
    ap.add_argument("--transformer-only", action="store_true",
                    help="BASELINE configs[1]: CLIP + sampling loop only (no decode / vocoder); not the default metric")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the BASELINE configs[4] leg (the training iteration from the reference's batch, B = 20 per GPU, 19 "
                         "layers: mel + captions -> BPE + CLIP + VQ encode -> loss + hand-written backward + all-reduce + clip + "
                         "AdamW + EMA; 5 + --train-steps iterations AFTER the sampling measurement, outside its timed region), "
                         "reported as the \"train\" object of the JSON line (text_to_sound_synthesis_amd/train_bench.py)")
    ap.add_argument("--no-train-prefetch", action="store_true",
                    help="training leg: run every batch's BPE / CLIP / VQ-encode prologue in line instead of prefetching the next "
                         "batch's on a side stream beside the replay in flight (GraphSolver.prefetch; +2.7 % measured)")
    ap.add_argument("--train-steps", type=int, default=200, help="timed iterations of the training leg (sustained rate; ~15 s)")
    ap.add_argument("--train-leg", action="store_true", help=argparse.SUPPRESS)   # (round-4 spelling; the leg is on by default)
    ap.add_argument("--collectives-selftest", action="store_true",
                    help="rendezvous + the path's collectives only (all-reduce of ones, caption scatter, waveform gather, one "
                         "gradient bucket) on stand-in tensors, no HIP kernels: backend nccl on GPUs, gloo without -- what "
                         "tests/test_shard_gloo.py drives through the self-launcher")
    ap.add_argument("--side-leg-limit", type=float, default=480.0,
                    help="N > 1: seconds the legs AFTER the timed region may take (roofline, stage split, communicator check, "
                         "training leg) before the line is printed without what is missing (HeadlineGuard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="print a per-stage split to stderr")
    ap.add_argument("--roofline-steps", type=int, default=12,
                    help="denoiser steps of the profiled roofline leg (HIP events around every GEMM launch)")
    ap.add_argument("--roofline-warm", type=int, default=10, help="un-profiled steps before them")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks ourselves, exactly the
    command the driver uses, and return its exit code (rank 0 of the child job prints the JSON line).  N > visible
    devices is refused loudly (exit code 2) -- except for --collectives-selftest on a box without GPUs (gloo)."""
    import subprocess
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not (args.collectives_selftest and n_dev == 0) and n_dev < args.gpus:
        print("bench.py: %d ranks requested (--gpus %d), %d device(s) visible -- refusing to measure fewer GPUs than asked for"
              % (args.gpus, args.gpus, n_dev), file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=env)


def world_from_env(args):
    """(world, rank, local_rank) of this process; exits when --gpus and the launcher's WORLD_SIZE disagree."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with --nproc-per-node %d (or leave WORLD_SIZE unset and let "
              "bench.py start the ranks)" % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def collectives_check(dev, world, rank):
    """What the N-rank line reports about its communicator: `rccl_ranks` = an all-reduce of ones over the job's process
    group (the ranks the collective actually saw), and one round of the path's own collectives on stand-in tensors --
    caption ids out (616 B per caption), waveforms back (868 KB per clip), one 48 MB gradient bucket -- timed."""
    import torch.distributed as dist
    from text_to_sound_synthesis_amd import shard
    out = {"backend": dist.get_backend() if world > 1 or dist.is_initialized() else None}

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    ones = torch.ones(1, device=dev)
    if dist.is_initialized():
        dist.all_reduce(ones)
    out["rccl_ranks"] = int(ones.item())
    n_total = 8 * world
    ids = torch.arange(n_total * 77, dtype=torch.long).view(n_total, 77) if rank == 0 else None
    force = dist.is_initialized()
    sync(); t0 = time.perf_counter()
    mine = shard.scatter_conditions(ids, n_total, (77,), dev, dtype=torch.long, always_collective=force)
    sync(); t1 = time.perf_counter()
    lo, hi = shard.shard_bounds(n_total, world, rank)
    assert mine.shape == (hi - lo, 77) and int(mine[0, 0]) == lo * 77, "scatter delivered the wrong slice"
    wave = mine[:, :1].float().expand(-1, 217088).contiguous().unsqueeze(1)
    sync(); t2 = time.perf_counter()
    allw = shard.gather_outputs(wave, n_total, always_collective=force)
    sync(); t3 = time.perf_counter()
    if rank == 0:
        assert allw.shape[0] == n_total and allw[:, 0, 0].tolist() == [float(i * 77) for i in range(n_total)], \
            "gather returned the clips out of caption order"
    red = shard.GradientReducer(bucket_bytes=48 << 20, always_collective=force)
    g = {"w%d" % i: torch.full((3 << 20,), float(rank + 1), device=dev) for i in range(5)}     # 5 x 12 MB: two buckets
    sync(); t4 = time.perf_counter()
    red.ready({k: g[k] for k in ("w0", "w1", "w2", "w3")})
    red.finish(g)
    sync(); t5 = time.perf_counter()
    want = sum(range(1, world + 1)) / world
    assert all(abs(float(v[0]) - want) < 1e-6 and abs(float(v[-1]) - want) < 1e-6 for v in g.values()), "gradient average"
    out.update(scatter_ms=round((t1 - t0) * 1e3, 3), gather_ms=round((t3 - t2) * 1e3, 3),
               gather_bytes=n_total * 217088 * 4, grad_bucket_ms=round((t5 - t4) * 1e3, 3), grad_bytes=5 * (3 << 20) * 4)
    return out


def selftest_main(args):
    """--collectives-selftest: the N-rank control flow without the HIP library (see parse())."""
    import torch.distributed as dist
    world, rank, local_rank = world_from_env(args)
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", local_rank) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("MASTER_ADDR"):
        dist.init_process_group("nccl" if cuda else "gloo", **({"device_id": dev} if cuda else {}))
    info = collectives_check(dev, world, rank)
    info.update(selftest=True, n_gpus=world, device=dev.type)
    if rank == 0:
        print(json.dumps(info))
    if dist.is_initialized():
        dist.destroy_process_group()


def cpu_baseline(n_layer, codes, T):
    """The path on this box's host cores: the unmodified reference under oracle/ref_harness.py
    when /root/reference exists (`kind: reference`, the build container), else the CPU oracle -- a restatement of the
    reference on the same torch-CPU ops (`kind: port`, the GPU box).  Bounded sample: torch thread count swept on one
    B=8 denoiser step (a B=1 step is GEMV-like and oversubscribes a many-core host) over every candidate up to the physical
    core count, then at the best count and for B in {1, 8}: 1 warm-up + 3 timed repeats (2 denoiser steps each), the MEDIAN
    per step, + decode + vocode, extrapolated to T steps.  Reports the better B; one full-length B=1 clip beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import diffsound_oracle as O
    import ref_harness as rh
    from text_to_sound_synthesis_amd import synth
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator
    torch.set_grad_enabled(False)
    use_ref = rh.available()
    if use_ref:
        ref = rh.build_dalle(n_layer=n_layer, diffusion_step=T, n_embed=codes)
        rdt = ref.transformer
        rdt.predict_start = ref.predict_start_with_truncation(rdt.predict_start, "top0.85r")
        rvoc = rh.build_vocoder()
    else:
        m = build_model(default_config(n_layer=n_layer, diffusion_step=T, n_embed=codes))
        synth.synth_init_(m, seed=0)
        sd = m.state_dict()
        g = Generator(80, 32, 3)
        synth.synth_init_(g, seed=0)
        gsd = g.state_dict()
        sched = O.make_schedule(T, codes + 1)

    def step_fn(B):
        cond = synth.synth_cond_emb(B, key="cpu.cond")
        u = synth.synth_uniform((B, codes + 1, 265), key="cpu.u")
        state = [O.initial_log_z(B, codes + 1)]

        def one(t):
            tt = torch.full((B,), t, dtype=torch.long)
            if use_ref:
                state[0] = rdt.p_sample(state[0], cond, tt)       # diffusion_transformer.py:353-357
            else:
                state[0] = O.p_sample_step(sd, sched, state[0], cond, tt, u)
        return one

    def tail(B):
        tok = synth.synth_tokens(B, 265, codes, 0.0, key="cpu.codes")
        t0 = time.perf_counter()
        mel = ref.decode_to_img(tok, (B, 256, 5, 53)) if use_ref else O.decode_tokens(sd, tok)
        t1 = time.perf_counter()
        if use_ref:
            rvoc((mel[:, 0] + 1) / 2)
        else:
            O.melgan_generator(gsd, O.mel_to_unit(mel[:, 0]))
        return t1 - t0, time.perf_counter() - t1

    hw = os.cpu_count() or 8
    phys = _physical_cores() or max(1, hw // 2)
    host = host_cpu_info()
    # what the host GIVES this process: a cgroup CPU quota or an affinity mask below the core count makes every thread count
    # above it slower, not faster (round 5's 128-core box: 1.65 s per step at 16 threads, 9.99 s at 128) -- sweep up to it
    usable = min(x for x in (phys, host["affinity_cpus"], host["cgroup_quota_cpus"]) if x)
    default_threads = torch.get_num_threads()
    # thread count swept UP TO the physical core count, every point measured -- no early exit -- on
    # one B=8 denoiser step after one warm-up step
    # (the hardware-thread count itself is not swept: with SMT siblings oversubscribed one B=8 step took 159 s on the 128-core
    #  box of round 5 against 1.3 s at the best count -- two such steps are the whole budget of this leg many times over)
    cands = sorted({n for n in (4, 8, 16, 32, 64, usable) if 1 <= n <= min(hw, max(usable, 8))})
    sweep = {}
    one = step_fn(8)
    for n in cands:
        torch.set_num_threads(n)
        one(T - 1)                                   # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        one(T - 2)
        sweep[n] = time.perf_counter() - t0
    best_n = min(sweep, key=sweep.get)
    torch.set_num_threads(best_n)
    med = lambda v: sorted(v)[len(v) // 2]
    res, reps = {}, {}
    for B in (1, 8):                                 # 1 warm-up + 3 timed repeats, median
        one = step_fn(B)
        one(T - 1)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(2):
                one(T - 2 - 2 * rep - i)
            ts.append((time.perf_counter() - t0) / 2)
        t_step = med(ts)
        tails = [tail(B) for _ in range(3 if B == 1 else 1)]      # B=8 decode + vocode is ~10 s: timed once (bounded sample)
        t_dec, t_voc = med([x[0] for x in tails]), med([x[1] for x in tails])
        res[B] = (B / (T * t_step + t_dec + t_voc), t_step, t_dec, t_voc)
        reps[B] = [round(x, 4) for x in ts]
    # one clip at FULL length (B = 1: all T steps, decode, vocode), when it fits the bounded sample: what the 2-step
    # extrapolation above is worth
    full = None
    if T * res[1][1] <= 40.0:
        one = step_fn(1)
        t0 = time.perf_counter()
        for t in range(T - 1, -1, -1):
            one(t)
        t_loop = time.perf_counter() - t0
        t_dec, t_voc = tail(1)
        full = {"seconds": round(t_loop + t_dec + t_voc, 2), "clips_per_s": round(1.0 / (t_loop + t_dec + t_voc), 5),
                "extrapolated_from_2_steps": round(res[1][0], 5)}
    torch.set_num_threads(default_threads)
    bB = max(res, key=lambda k: res[k][0])
    value = res[bB][0]
    if full is not None and full["clips_per_s"] > value:       # the CPU gets its best measured rate
        value = full["clips_per_s"]
    return {"value": value, "unit": "clips/s", "cores": best_n, "kind": "reference" if use_ref else "port",
            "host_hw_threads": hw, "host_physical_cores": phys, "host": host, "usable_cpus": usable,
            "protocol": "BASELINE.md section 4: B in {1, 8}, 1 warm-up + 3 timed repeats, median; value = the better B",
            "thread_sweep_s_per_B8_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "full_length_B1": full,
            "per_batch": {"B=%d" % B: {"clips_per_s": round(v[0], 5), "s_per_denoiser_step_median": round(v[1], 3),
                                        "s_per_denoiser_step_repeats": reps[B], "s_decode": round(v[2], 2),
                                        "s_vocode": round(v[3], 2)} for B, v in res.items()},
            "sample": "%s, fp32 torch-CPU, %d torch threads (best of the sweep %s on a B=8 step; %d physical cores, %d hardware "
                      "threads), B=%d: median of 3 repeats of 2 of the %d denoiser steps (%.3f s per step) + decode (%.2f s) + "
                      "vocode (%.2f s), extrapolated to %d steps%s"
                      % ("the unmodified reference under oracle/ref_harness.py" if use_ref else
                         "CPU oracle (restatement of the reference; /root/reference is not on this box)",
                         best_n, sorted(sweep), phys, hw, bB, T, res[bB][1], res[bB][2], res[bB][3], T,
                         "; one full-length B=1 clip measured beside it: %.2f s" % full["seconds"] if full else "")}


def host_cpu_info():
    """What the box gives this process: CPU model (BASELINE.md section 4 asks for it), the scheduler affinity, the cgroup
    CPU quota (v2 cpu.max, else v1 cfs quota / period) as a CPU count (None = unlimited), NUMA nodes of the affinity mask."""
    info = {"model": None, "affinity_cpus": None, "cgroup_cpu_max": None, "cgroup_quota_cpus": None, "numa_nodes_online": None}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    info["cgroup_cpu_max"], info["cgroup_quota_cpus"] = _cgroup_quota()
    try:
        with open("/sys/devices/system/node/online") as f:
            info["numa_nodes_online"] = f.read().strip()
    except OSError:
        pass
    return info


def _cgroup_quota(root="/sys/fs/cgroup"):
    """(raw text, quota in CPUs rounded up or None)"""
    import math
    try:
        with open(os.path.join(root, "cpu.max")) as f:                     # cgroup v2: "<quota|max> <period>"
            raw = f.read().strip()
        q, per = raw.split()
        return raw, (None if q == "max" else max(1, math.ceil(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            q = int(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            per = int(f.read())
        return "%d %d" % (q, per), (None if q <= 0 else max(1, math.ceil(q / per)))
    except (OSError, ValueError):
        return None, None


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; None when the file does not say."""
    try:
        cores, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cores.add((phys, line.split(":")[1].strip()))
        return len(cores) or None
    except OSError:
        return None


def pmc_traffic(kernel, template_tail, algorithmic_mb):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary over denoiser sampling steps
    at B=64 (profiles/r*_pmc_denoiser_step_b64.json, made on the GPU box by tools/profile_round.sh: tools/pmc_step.py under
    separate --pmc passes, tools/pmc_summarize.py with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).
    PMC counters cannot be read inside this process, so the number is a committed measurement -- reported ONLY while
    (a) the kernel sources still hash to what was measured (`_meta.source_sha16`) and (b) the summary lists the very
    instantiations this run timed: the FULL symbol must match, template arguments included (`template_tail`, e.g.
    ",true>" for the 272-row program of the padded-row mode).  Otherwise traffic is null and the note says why."""
    import glob
    from text_to_sound_synthesis_amd.build import source_fingerprint
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_denoiser_step_b64.json")))
    if not files:
        return None, "no PMC summary under profiles/"
    path = files[-1]
    table = json.load(open(path))
    meta = table.get("_meta", {})
    now = source_fingerprint()
    if meta.get("source_sha16") != now:
        return None, ("stale: %s was measured on kernel sources %s, the tree is %s -- re-run tools/profile_round.sh"
                      % (os.path.basename(path), meta.get("source_sha16", "(unrecorded)"), now))
    want = kernel.split(" (")[0].replace(" ", "")
    hit = {name: r for name, r in table.items() if name != "_meta" and name.replace(" ", "").startswith(want + "<")
           and name.replace(" ", "").endswith(template_tail) and r.get("hbm_read_MB_per_launch") is not None}
    if not hit:
        return None, ("%s lists no %s<...%s instantiation (it has: %s)"
                      % (os.path.basename(path), want, template_tail, meta.get("gemm_instantiations", "?")))
    n = sum(r["dispatches"] for r in hit.values())
    rd = sum(r["hbm_read_MB_per_launch"] * r["dispatches"] for r in hit.values()) / n
    wr = sum((r["hbm_write_MB_per_launch"] or 0.0) * r["dispatches"] for r in hit.values()) / n
    l2 = sum(r["l2_hit_rate"] * r["dispatches"] for r in hit.values()) / n
    return round((rd + wr) * 1e6), (
        "bytes per launch = %.0f MB read (2 x FETCH_SIZE; Infinity-Cache hits included) + %.0f MB written (WRITE_SIZE), mean "
        "of %d dispatches of %s, L2 hit rate %.2f; algorithmic operand + residual + result bytes of the same launches average "
        "%.0f MB -> traffic / algorithmic = %.2f; source profiles/%s (kernel sources %s)"
        % (rd, wr, n, sorted(hit), l2, algorithmic_mb, (rd + wr) / algorithmic_mb, os.path.basename(path), now))


def gemm_algorithmic_mb(B, rows, D=1024, mlp=4):
    """Mean over the six GEMM launches of a block of operand + residual + result bytes (packed fp16 planes = 4 bytes per
    element, like fp32): what one launch of the per-sample program has to move at least, in MB."""
    M = B * rows
    a, o = M * D * 4.0, M * D * 4.0
    qkv = a + 3 * D * D * 4.0 + 3 * o
    proj = a + D * D * 4.0 + 2 * o                      # result + residual
    crossq = a + D * D * 4.0 + o
    fc1 = a + mlp * D * D * 4.0 + mlp * o
    fc2 = mlp * a + mlp * D * D * 4.0 + 2 * o
    return (qkv + 2 * proj + crossq + fc1 + fc2) / 6.0 / 1e6


def timed_loop(one_step, warmup, steps, device, world, info=None):
    """W untimed warm-up steps, then EXACTLY K timed steps bracketed by device synchronisation + a barrier on both
    sides; returns (elapsed seconds = MAX over ranks, last step's output).  Shared by main() and the world-size-2
    gloo test of the control flow (tests/test_shard_gloo.py).  `info` (a dict) receives the spread over the ranks of
    each rank's OWN time to finish its K steps (before the closing barrier): {"rank_s_min", "rank_s_max"}."""
    import torch.distributed as dist

    def sync_all():
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        one_step()
    sync_all()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = one_step()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    own = time.perf_counter() - t0                  # this rank's K steps, before it waits for the others
    sync_all()
    elapsed = time.perf_counter() - t0
    lo = hi = own
    if world > 1:
        te = torch.tensor([elapsed, own, -own], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed, hi, lo = float(te[0].item()), float(te[1].item()), -float(te[2].item())
    if info is not None:
        info.update(rank_s_min=lo, rank_s_max=hi)
    return elapsed, out


class HeadlineGuard:
    """The measurement is over when timed_loop returns; everything after it (profiled roofline leg, the stage split, the
    communicator check, the training leg, the CPU baseline) is a side leg and must never cost the line.  Started on EVERY
    rank right after the timed region: if the side legs have not finished `limit_s` later -- a rank died inside a leg's
    collective and the others wait for it -- `emit(reason)` prints the line as far as it got (rank 0) and the process
    exits 0 without tearing the process group down.  finish() is the normal path: True = the caller prints the line."""

    def __init__(self, limit_s, emit, exit_fn=None):
        import threading
        self.limit_s, self.emit, self.exit_fn = limit_s, emit, exit_fn or os._exit
        self.lock, self.closed = threading.Lock(), False
        self.timer = threading.Timer(limit_s, self._fire)
        self.timer.daemon = True

    def start(self):
        self.timer.start()
        return self

    def _fire(self):
        with self.lock:
            if self.closed:
                return
            self.closed = True
            try:
                print("bench.py: side legs unfinished after %d s -- printing the measured line with \"incomplete\" and ending the "
                      "process (the timed region had completed)" % self.limit_s, file=sys.stderr)
                self.emit("side legs unfinished after %d s" % self.limit_s)
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                self.exit_fn(0)

    def finish(self):
        with self.lock:
            if self.closed:            # the timer is printing / has printed: it ends the process
                return False
            self.closed = True
        self.timer.cancel()
        return True


def run_side_legs(legs, world, rank, line):
    """legs = [(name, fn, runs_on_this_rank, contains_collectives)], in order.  An exception inside a leg is filed as
    line[name] = {"error": ...} (rank 0) and reported on stderr instead of propagating.  With N ranks a rank that left a
    COLLECTIVE leg early must not enter another collective (the others are still inside the failed one): every later
    collective leg is skipped on it and False is returned -- the caller then prints and exits without tearing the process
    group down; the other ranks are ended by their HeadlineGuard."""
    in_step = True
    for name, fn, mine, collective in legs:
        if not mine or (collective and not in_step):
            continue
        try:
            fn()
        except Exception as e:
            if rank == 0:
                line[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench.py: side leg %r failed on rank %d: %s: %s" % (name, rank, type(e).__name__, e), file=sys.stderr)
            if collective and world > 1:
                in_step = False
    return in_step


def result_line(args, world, elapsed, n_total):
    """The driver's one-line JSON contract (whole-job aggregate over all ranks)."""
    value = n_total * args.steps / elapsed
    T, B = args.diffusion_steps, args.batch
    return {
        "metric": "10s clips/sec whole-node, 100-step Diffsound sample + VQ decode + vocoder",
        "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"fp32": "f32",
                  "f16x2": "f32 via 2-way fp16 split (3 MFMA passes, fp32 accumulate)"}[args.precision],
        "data": "synthetic captions (5-15 words, BPE-tokenised in the timed region) + seeded random-init weights of the "
                "reference's shapes",
        "config": {"workload": ("BASELINE configs[1]: transformer only, batch %d per GPU, %d diffusion steps, codebook %d: "
                                "CLIP text tower -> 19-layer denoiser + sampler (tokens; no decode / vocoder)"
                                if getattr(args, "transformer_only", False) else
                                "BASELINE configs[2]: full pipeline, batch %d per GPU, %d diffusion steps, "
                                "codebook %d: CLIP text tower -> 19-layer denoiser -> SpecVQGAN decode -> MelGAN "
                                "22 kHz") % (B, T, args.codes),
                   "global_batch": n_total, "n_layer": args.n_layer, "parallelism": "caption-sharded x%d" % world,
                   "noise": ("in-kernel Philox4x32-10 keyed by (seed, global caption id, step, position, class)"
                             if getattr(args, "rng", "philox") == "philox" else "torch.rand per step"),
                   "denoiser_tflops_effective": round(value * GFLOP_PER_SAMPLE_STEP * T / 1e3, 2)},
    }


def main():
    # dmabuf IPC.  The host driver of this pool only supports dmabuf handles: with the legacy IPC mode RCCL's intra-node
    # transport setup (and any CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument` as soon
    # as two ranks exchange buffer handles -- an N-rank job then dies in its first collective.  Set BEFORE the first HIP call
    # and on every entry path: the self-launcher's children inherit it, ranks started by the driver's own
    # `torch.distributed.run ... bench.py --gpus N` get it here.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # not under a launcher: start the ranks ourselves
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.collectives_selftest:
        return selftest_main(args)
    world, rank, local_rank = world_from_env(args)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: local rank %d but %d device(s) visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.set_grad_enabled(False)

    from text_to_sound_synthesis_amd import _lib, shard, synth
    from text_to_sound_synthesis_amd.config import build_model, default_config
    from text_to_sound_synthesis_amd.modeling.vocoder import Generator

    T, B = args.diffusion_steps, args.batch
    model = build_model(default_config(n_layer=args.n_layer, diffusion_step=T, n_embed=args.codes, with_clip=True))
    synth.synth_init_(model, seed=0)
    model = model.to(dev).eval()
    voc = synth.synth_init_(Generator(80, 32, 3), seed=0).to(dev).eval()
    dt = model.transformer
    dt.truncation_r = 0.85
    dt.transformer.precision = args.precision
    n_total = B * world
    # Every rank holds the caption list -- synthetic caption STRINGS; the reference's sharded sampler
    # opens the same dataset on every rank, Codebook/evaluation/generate_samples_caps.py:147-153 --, rank 0 scatters caption
    # INDICES, and each rank tokenises ITS captions inside the timed region (shard.scatter_captions; rank 0's host work does
    # not grow with N) with the package's BPE tokenizer (clip.tokenize semantics: <SOT> word pieces <EOT>, context 77) on
    # the closed-vocabulary merge table text-to-sound-synthesis_amd/data/bpe_closed_vocab.json (tokenizer.CLOSED_VOCAB_PATH)
    # -- the part of CLIP's table these captions exercise, ids checked against the reference's tokenizer when the file was
    # made (the 1.3 MB full table is not on the GPU box) --, then runs the CLIP text tower on them.
    from text_to_sound_synthesis_amd import tokenizer as tz
    captions = synth.synth_captions(n_total, seed=7)
    bpe = tz.SimpleTokenizer(bpe_path=tz.CLOSED_VOCAB_PATH)
    bpe_ids = lambda strings: tz.tokenize(strings, context_length=77, add_start_and_end=True, tokenizer=bpe)["token"]
    torch.manual_seed(1234 + rank)
    lo_id, hi_id = shard.shard_bounds(n_total, world, rank)
    my_ids = torch.arange(lo_id, hi_id, device=dev, dtype=torch.long)
    stage = {"scatter": 0.0, "kv": 0.0, "sample": 0.0, "decode": 0.0, "vocode": 0.0, "gather": 0.0}

    def one_step(timed_stages=False):
        def mark():
            if timed_stages:
                torch.cuda.synchronize()
            return time.perf_counter()
        t0 = mark()
        toks, _ = shard.scatter_captions(captions, n_total, dev, bpe_ids)
        t1 = mark()
        # per-caption in-kernel noise keyed by the GLOBAL caption index: a caption's clip is the same at every world size
        out = dt.sample(condition_token=toks, condition_mask=None, condition_embed=None, filter_ratio=0,
                        caption_ids=None if args.rng == "torch" else my_ids, seed=1234)
        t2 = mark()
        if args.transformer_only:
            tok = out["content_token"]
            allt = shard.gather_outputs(tok, n_total)
            return allt if allt is not None else tok
        mel = model.decode_to_img(out["content_token"], (B, 256, 5, 53))
        t3 = mark()
        wave = voc(mel[:, 0], scale=0.5, shift=0.5)
        t4 = mark()
        allw = shard.gather_outputs(wave, n_total)
        t5 = mark()
        if timed_stages:
            for k, v in zip(("scatter", "sample", "decode", "vocode", "gather"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                stage[k] += v
        return allw if allw is not None else wave

    spread = {}
    elapsed, w = timed_loop(one_step, args.warmup, args.steps, dev, world, info=spread)
    if args.transformer_only:
        assert w.shape[-1] == 265 and int(w.max()) < args.codes       # no [MASK] left at t = 0
    else:
        assert torch.isfinite(w).all() and w.shape[-1] == 217088

    # ---- the measurement is done: from here on only side legs, each of which may fail without costing the line ----------
    line = result_line(args, world, elapsed, n_total) if rank == 0 else {}
    bare = json.dumps(dict(line, incomplete="side legs did not finish"))

    def emit(reason=None):
        if rank != 0:
            return
        try:
            text = json.dumps(dict(line, incomplete=reason) if reason else line)
        except Exception:                       # (the timer thread caught the dict mid-update)
            text = bare
        print(text, flush=True)
    guard = HeadlineGuard(args.side_leg_limit, emit)
    if rank == 0:
        # a rank that DIES inside a side leg makes torch.distributed.run SIGTERM the others long before the guard's timer: the
        # measurement is complete at this point, so rank 0 prints it on the way out (exit code 3 = line printed, legs cut short)
        import signal

        def on_term(signum, frame):
            if guard.finish():
                emit("SIGTERM during the side legs (another rank exited)")
                sys.stdout.flush()
            os._exit(3)
        signal.signal(signal.SIGTERM, on_term)
    if world > 1:           # one rank cannot wait for another: its legs end by themselves (cpu_baseline alone takes minutes)
        guard.start()

    def leg_host_copy():
        # what handing the result over to the host costs (generate_sample does it before writing .wav files): the batch's
        # waveforms HBM -> pinned host memory, timed on its own -- reported beside `value`, never part of it
        host = torch.empty(w.shape, dtype=w.dtype, pin_memory=True)
        host.copy_(w, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            host.copy_(w, non_blocking=True)
        torch.cuda.synchronize()
        copy_s = (time.perf_counter() - t0) / 5
        line["host_copy"] = {"bytes_per_step": w.numel() * w.element_size(), "ms_per_step": round(copy_s * 1e3, 3),
                             "GB_per_s": round(w.numel() * w.element_size() / copy_s / 1e9, 1),
                             "clips_per_s_incl_copy": round(n_total * args.steps / (elapsed + copy_s * args.steps), 4)}

    def leg_roofline():
        # profiled legs of a few denoiser steps: HIP events around every GEMM launch (ds_profile_*)
        L = _lib.lib()
        cond = synth.synth_cond_emb(B, key="bench.cond").to(dev)
        # (kernel-symbol prefix, passes of MFMA work per algorithmic flop, MFMA peak of the operand type)
        KIND = {"fp32": ("ds_gemm_kernel<%d,%d,0,0>", 1, PEAK_F32_MFMA_TFLOPS, "fp32 MFMA 32x32x2"),
                "f16x2": ("ds_gemm_f16x2_kernel<%d,%d>", 3, PEAK_16BIT_MFMA_TFLOPS, "3-pass fp16 MFMA 32x32x16")}

        def leg(precision, warm=None, steps=None):
            warm = args.roofline_warm if warm is None else warm
            steps = args.roofline_steps if steps is None else steps
            dt.transformer.precision = precision
            kv = dt.transformer.condition_kv(cond, dt._schedule_table())
            x = torch.full((B, 265), args.codes, device=dev, dtype=torch.long)
            u = torch.rand((B, args.codes + 1, 265), device=dev)
            t = torch.full((B,), T - 1, device=dev, dtype=torch.long)
            x = dt.p_sample_tokens(x, kv, t, u, initial=True)            # warm-up (also builds the pack)
            for i in range(warm):                                        # bring the board to the timed loop's thermal state
                x = dt.p_sample_tokens(x, kv, t, u, initial=False)
            torch.cuda.synchronize()
            L.ds_profile_enable(1)
            for i in range(steps):
                t = torch.full((B,), max(T - 2 - i, 0), device=dev, dtype=torch.long)
                x = dt.p_sample_tokens(x, kv, t, u, initial=False)
            L.ds_profile_enable(0)
            ms, fl, n = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
            _lib.check(L.ds_profile_collect_n(ms, fl, n, 5))
            fmt, passes, mfma_peak, what = KIND[precision]
            names = [fmt % bb for bb in ((128, 128), (128, 64), (64, 64))] + ["(unused)", "(unused)"]
            if precision == "f16x2":   # packed-operand launches of the 128x128 config go through the balanced kernel
                names[0] = "ds_gemm_f16x2_hybrid_kernel (128x128 tiles + 64x64 tail tiles)"
                names[3] = "ds_gemm_f16x2_ps_kernel (per-sample 272x256 tiles, 8-phase ping-pong main loop)"
                names[4] = "ds_gemm_f16x2_ph_kernel (half-sample 144|128x256 tiles, 3 stages, 2 phases per k-tile)"
            dom = max(range(5), key=lambda c: ms[c])   # the kernel symbol with the largest total time
            ach = fl[dom] / (ms[dom] * 1e-3) / 1e12
            peak = mfma_peak / passes                  # ceiling in algorithmic (2MNK) flops of this formulation
            # the committed PMC passes ran the default f16x2 step at B=64; other legs / sizes have no measurement
            measured = precision == "f16x2" and dom == 3 and (B, args.n_layer, args.codes) == (64, 19, 256)
            traffic, traffic_note = None, None
            if measured:
                h = dt.transformer.packed(dt._schedule_table())["handle"]
                rows = L.ds_denoiser_rows_per_sample(h, B)
                traffic, traffic_note = pmc_traffic(names[dom], ",true>" if rows == 272 else ",false>",
                                                    gemm_algorithmic_mb(B, rows))
            return {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": "%s (%s, dense loader)" % (names[dom], what), "launches": int(n[dom]),
                    "avg_launch_us": round(ms[dom] * 1e3 / max(1, n[dom]), 2),
                    "avg_launch_gflop": round(fl[dom] / max(1, n[dom]) / 1e9, 3),
                    "mfma_passes_per_flop": passes, "mfma_peak": mfma_peak,
                    "all_gemm_tiles": {names[c]: {"launches": int(n[c]),
                                                  "avg_launch_us": round(ms[c] * 1e3 / max(1, n[c]), 2),
                                                  "tflops": round(fl[c] / max(ms[c], 1e-9) / 1e9, 2)}
                                       for c in range(5) if n[c]},
                    "all_gemm_tflops": round(sum(fl) / (sum(ms) * 1e-3) / 1e12, 2)}
        try:
            line["roofline"] = leg(args.precision)
            if args.precision != "fp32":               # the exact-fp32 MFMA kernel on the same shapes, for reference
                line["roofline_fp32_mfma_kernel"] = leg("fp32", warm=0, steps=3)    # (a reference point: three steps suffice)
        finally:
            L.ds_profile_enable(0)
            dt.transformer.precision = args.precision

    def leg_stage_split():
        # one more step with a synchronisation between the stages (outside the timed region): where a batch's time goes.
        # EVERY rank runs it (the step contains the scatter / gather collectives); rank 0 reports its own split and, for
        # N > 1, the slowest rank's per stage
        one_step(timed_stages=True)
        names = [k for k in stage if k != "kv"]
        mine = torch.tensor([stage[k] for k in names], device=dev, dtype=torch.float64)
        worst = mine.clone()
        if world > 1:
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        split = {k: round(float(v) * 1e3, 2) for k, v in zip(names, mine.tolist())}
        if world > 1:
            split["max_over_ranks"] = {k: round(float(v) * 1e3, 2) for k, v in zip(names, worst.tolist())}
        split["note"] = "one extra step, device-synchronised between stages: tokenise+scatter | CLIP + 100-step sampling | SpecVQGAN decode | MelGAN vocode | gather"
        if rank == 0:
            line["stage_ms"] = split
        if args.stage_times and rank == 0:
            print("stage seconds (1 step, B=%d): %s" % (B, {k: round(v, 3) for k, v in stage.items()}), file=sys.stderr)

    def leg_communicator():
        # the communicator this line was measured on: ranks an all-reduce saw, the path's collectives on stand-in tensors
        info = collectives_check(dev, world, rank)
        info["ms_per_step_per_rank"] = {"min": round(spread["rank_s_min"] / args.steps * 1e3, 2),
                                        "max": round(spread["rank_s_max"] / args.steps * 1e3, 2)}
        if rank == 0:
            line["rccl"] = info

    def leg_train():    # every rank takes part (data-parallel step; one GPU: the iteration replayed as one hipGraph)
        nonlocal model, voc, dt
        model = voc = dt = None                   # the sampling model makes room for the training step's activations
        torch.cuda.empty_cache()
        from text_to_sound_synthesis_amd import train_bench
        torch.set_grad_enabled(False)
        r = train_bench.run(batch=20, steps=args.train_steps, warmup=5, n_layer=args.n_layer, codes=args.codes,
                            precision="f16x2", graph=True, world=world, rank=rank, dev=dev, prefetch=not args.no_train_prefetch)
        if rank == 0:
            line["train"] = {"it_per_s_sustained": round(r["it_per_s_sustained"], 3), "it_per_s_replay": round(r["it_per_s_replay"], 3),
                             "iterations": r["steps"], "recaptures": r["recaptures"], "recapture_reasons": r["recapture_reasons"],
                             "monitor_log2": r["monitor_log2"], "loss_scale_exp": r["loss_scale_exp"],
                             "samples_per_s": round(r["samples_per_s"], 2), "ms_per_it": round(r["ms_per_step"], 2),
                             "ms_per_replay_median": round(r["ms_per_replay_median"], 2),
                             "ms_per_iteration_max": round(r["ms_per_iteration_max"], 2),
                             "batch_per_gpu": 20, "graph": r["graph"], "prefetch": r["prefetch"], "dtype": r["dtype"],
                             "loss": r["loss"], "grad_norm": r["grad_norm"], "workload": r["config"]["workload"],
                             "parallelism": r["config"]["parallelism"]}

    def leg_cpu_baseline():
        line["cpu_baseline"] = cpu_baseline(args.n_layer, args.codes, T)

    # (name the error is filed under, the leg, who runs it, whether it contains collectives)
    legs = [("host_copy", leg_host_copy, rank == 0 and world == 1 and not args.transformer_only, False),
            ("roofline", leg_roofline, rank == 0 and not args.no_roofline, False),
            ("stage_ms", leg_stage_split, not args.transformer_only, True),
            ("rccl", leg_communicator, world > 1, True),
            ("train", leg_train, not args.no_train_leg, True),
            ("cpu_baseline", leg_cpu_baseline, rank == 0 and world == 1 and not args.no_cpu_baseline, False)]
    in_step = run_side_legs(legs, world, rank, line)
    if not guard.finish():
        time.sleep(3600)                          # the guard's timer is printing the line and ends the process
    emit()
    if world > 1 and in_step:
        dist.destroy_process_group()
    elif world > 1:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)                               # the others wait inside a collective until their own guards end them


if __name__ == "__main__":
    main()
