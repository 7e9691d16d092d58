"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the caption sharding used by bench.py /
the pipeline: scatter the caption conditioning from rank 0, compute locally, gather in caption order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from text_to_sound_synthesis_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        cond_all = torch.arange(n_items * 6, dtype=torch.float32).view(n_items, 2, 3) if rank == 0 else None
        mine = shard.scatter_conditions(cond_all, n_items, (2, 3), torch.device("cpu"))
        lo, hi = shard.shard_bounds(n_items, world, rank)
        assert mine.shape == (hi - lo, 2, 3)
        # stand-in for the per-caption pipeline: a function of the caption's conditioning and of
        # noise keyed by the GLOBAL caption index (independent of rank / batch position)
        noise = shard.per_caption_noise(range(lo, hi), step=7, shape_tail=(5, 3), device=torch.device("cpu")).flatten(1)
        local = mine.sum((1, 2))[:, None] + noise
        out = shard.gather_outputs(local, n_items)
        if rank == 0:
            q.put(out.numpy().copy())            # numpy: pickled by value (a tensor travels as a shared fd the exiting worker may close)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 5), (3, 7)])
def test_scatter_compute_gather(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cond_all = torch.arange(n_items * 6, dtype=torch.float32).view(n_items, 2, 3)
    noise = shard.per_caption_noise(range(n_items), step=7, shape_tail=(5, 3), device=torch.device("cpu")).flatten(1)
    want = cond_all.sum((1, 2))[:, None] + noise          # what a single process would produce
    assert torch.equal(torch.from_numpy(out), want)


def _caption_worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from text_to_sound_synthesis_amd import synth, tokenizer as tz
        caps = synth.synth_captions(n_items + 3, seed=7)          # every rank holds the list (the dataset)
        bpe = tz.SimpleTokenizer(bpe_path=tz.CLOSED_VOCAB_PATH)
        tok = lambda strings: tz.tokenize(strings, context_length=77, add_start_and_end=True, tokenizer=bpe)["token"]
        order = list(range(n_items + 2, 2, -1)) if rank == 0 else None     # rank 0 alone knows which captions run, and in which order
        ids, mine = shard.scatter_captions(caps, n_items, torch.device("cpu"), tok, order=order)
        lo, hi = shard.shard_bounds(n_items, world, rank)
        assert ids.shape == (hi - lo, 77) and ids.dtype == torch.long and len(mine) == hi - lo
        out = shard.gather_outputs(ids, n_items)
        if rank == 0:
            q.put((out.numpy().copy(), tok([caps[i] for i in order]).numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 8), (3, 7)])
def test_per_rank_tokenisation_equals_rank0_tokenise_then_scatter(world, n_items):
    """each rank tokenises its OWN captions (rank 0 scatters caption indices, 8 B each); gathered back in
    caption order the ids are exactly what rank 0 would have produced by tokenising everything and scattering the ids."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_caption_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.dtype == want.dtype and (got == want).all()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _bench_worker(rank, world, port, q):
    """bench.py's timed loop and JSON line under a world-size-2 gloo group: caption scatter -> (stand-in) local
    pipeline whose speed differs per rank -> waveform gather, exactly the control flow main() runs per step."""
    import importlib.util
    import time
    from types import SimpleNamespace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        dev = torch.device("cpu")
        B, n_total = 3, 3 * world
        tok_all = torch.arange(n_total * 77).view(n_total, 77) if rank == 0 else None
        calls = []

        def one_step():
            toks = shard.scatter_conditions(tok_all, n_total, (77,), dev, dtype=torch.long)
            time.sleep(0.05 * (rank + 1))                     # rank 1 is the slow one
            wave = toks[:, :1].float().expand(-1, 16).contiguous()
            calls.append(1)
            return shard.gather_outputs(wave, n_total)
        elapsed, out = bench.timed_loop(one_step, warmup=1, steps=3, device=dev, world=world)
        assert len(calls) == 4                                # W + K steps, no more
        assert elapsed >= 3 * 0.05 * world                    # the MAX over ranks: bounded below by the slowest rank
        args = SimpleNamespace(steps=3, warmup=1, batch=B, diffusion_steps=100, n_layer=19, codes=256, precision="f16x2")
        line = bench.result_line(args, world, elapsed, n_total) if rank == 0 else None
        q.put((rank, elapsed, line, None if out is None else out[:, 0].tolist()))
    finally:
        dist.destroy_process_group()


def test_bench_timed_loop_and_json_line_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r[0], r[1:]) for r in (q.get(timeout=180), q.get(timeout=180)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (e0, line, out0), (e1, _, out1) = got[0], got[1]
    assert e0 == e1                                           # every rank holds the same, all-reduced time
    assert out1 is None and out0 == [float(i * 77) for i in range(6)]   # gathered in caption order on rank 0
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in line
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["global_batch"] == 6
    assert abs(line["value"] - 6 * 3 / e0) < 1e-3             # whole-job aggregate: all captions of all ranks / time
    assert abs(line["ms_per_step"] - e0 / 3 * 1e3) < 1e-2


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        grads = {"b.weight": torch.randn(300, 7, generator=g), "a.bias": torch.randn(11, generator=g),
                 "c.emb": torch.randn(64, 33, generator=g)}
        shard.allreduce_gradients(grads, bucket_bytes=4096)        # small buckets: several flushes
        # plain numpy through the queue: a torch tensor travels as a shared file descriptor, which the parent can only
        # rebuild while this process is still alive (the test used to race the worker's exit)
        q.put((rank, {k: v.numpy().copy() for k, v in grads.items()}))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """The data-parallel gradient reduction of the training step: bucketed all-reduce + average, every rank ends up
    with the mean of the per-rank gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank = []
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        per_rank.append({"b.weight": torch.randn(300, 7, generator=g), "a.bias": torch.randn(11, generator=g),
                         "c.emb": torch.randn(64, 33, generator=g)})
    got = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in got.items()}
    for k in per_rank[0]:
        mean = (per_rank[0][k] + per_rank[1][k]) / 2
        assert torch.allclose(got[0][k], mean, atol=1e-6) and torch.equal(got[0][k], got[1][k])



class _LinStep:
    """TrainStep stand-in (CPU): y = x W^T, loss = mean over the batch of sum(y^2); plain SGD-free AdamW written out."""

    def __init__(self, w):
        self.w = w

    def loss_and_grads(self, x):
        y = x @ self.w.T
        return (y * y).sum() / x.shape[0], {"w": (2.0 / x.shape[0]) * y.T @ x}

    def adamw_step(self, grads, state, step, lr, betas, eps, weight_decay):
        g = grads["w"]
        if "w" not in state:
            state["w"] = (torch.zeros_like(g), torch.zeros_like(g))
        m, v = state["w"]
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1, bc2s = 1 - betas[0] ** step, (1 - betas[1] ** step) ** 0.5
        self.w.mul_(1 - lr * weight_decay).sub_((lr / bc1) * m / (v.sqrt() / bc2s + eps))


def _solver_run(xs, allreduce):
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, Solver
    g = torch.Generator().manual_seed(5)
    w = torch.randn(3, 4, generator=g)
    s = Solver(_LinStep(w), lr=1e-2, clip_grad_norm=GradClipWindow(max_norm=0.5), allreduce=allreduce)
    for x in xs:
        s.step(x)
    return w


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(9)
        batches = [torch.randn(8, 4, generator=g) for _ in range(4)]
        lo, hi = shard.shard_bounds(8, world, rank)
        q.put((rank, _solver_run([b[lo:hi] for b in batches], shard.allreduce_gradients).numpy().copy()))   # numpy: no fd sharing
    finally:
        dist.destroy_process_group()


def test_data_parallel_solver_equals_single_process():
    """BASELINE config 4's shape on CPU: two ranks, each with half of every batch, gradients averaged by the bucketed
    all-reduce before the clip -> both ranks hold the weights a single process gets from the whole batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(9)
    batches = [torch.randn(8, 4, generator=g) for _ in range(4)]
    single = _solver_run(batches, None)
    got = {r: torch.from_numpy(v) for r, v in got.items()}
    assert torch.equal(got[0], got[1])
    assert torch.allclose(got[0], single, rtol=1e-5, atol=1e-7)


class _BlockStep:
    """TrainStep stand-in with the REAL hand-over protocol: three 'blocks' y = W_k x whose weight gradients are handed to
    on_grads as the (reversed) backward produces them, plus bias gradients that are only final at the end (like the
    biases / norm gains TrainStep un-scales in one multiply after the loop)."""

    def __init__(self, ws, bs):
        self.ws, self.bs = ws, bs
        self.handed = []

    def loss_and_grads(self, x, on_grads=None):
        hs = [x]
        for w, b in zip(self.ws, self.bs):
            hs.append(hs[-1] @ w.T + b)
        loss = (hs[-1] ** 2).sum() / x.shape[0]
        d = (2.0 / x.shape[0]) * hs[-1]
        g = {}
        for k in reversed(range(len(self.ws))):
            g["w%d" % k] = d.T @ hs[k]
            g["b%d" % k] = d.sum(0) * 8.0          # still "scaled": un-scaled below, after every hand-over
            if on_grads is not None:
                on_grads({"w%d" % k: g["w%d" % k]}, ())
                self.handed.append("w%d" % k)
            d = d @ self.ws[k]
        for k in range(len(self.ws)):
            g["b%d" % k].mul_(1.0 / 8.0)
        return loss, g

    def adamw_step(self, grads, state, step, lr, betas, eps, weight_decay):
        for k, w in enumerate(self.ws):
            w.sub_(lr * grads["w%d" % k])
            self.bs[k].sub_(lr * grads["b%d" % k])


def _block_model():
    g = torch.Generator().manual_seed(21)
    return [torch.randn(6, 6, generator=g) * 0.3 for _ in range(3)], [torch.randn(6, generator=g) * 0.1 for _ in range(3)]


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, Solver
        g = torch.Generator().manual_seed(33)
        batches = [torch.randn(8, 6, generator=g) for _ in range(3)]
        lo, hi = shard.shard_bounds(8, world, rank)
        out = {}
        for mode in ("overlapped", "after"):
            ws, bs = _block_model()
            step = _BlockStep(ws, bs)
            # 100-byte buckets: every weight gradient (144 bytes) flushes its own asynchronous all-reduce during the backward
            red = shard.GradientReducer(bucket_bytes=100) if mode == "overlapped" else None
            s = Solver(step, lr=1e-2, clip_grad_norm=GradClipWindow(max_norm=0.5),
                       allreduce=None if red is not None else shard.allreduce_gradients, reducer=red)
            for b in batches:
                s.step(b[lo:hi])
            out[mode] = ([w.numpy().copy() for w in ws], [b.numpy().copy() for b in bs])   # numpy: no fd sharing (see _grad_worker)
            if red is not None:
                assert step.handed[:3] == ["w2", "w1", "w0"] and not red._inflight and not red._done
        # a name handed over twice, or one that never reaches the final dict, is an error -- not a silent wrong reduction
        red = shard.GradientReducer(bucket_bytes=1 << 20)
        red.ready({"x": torch.ones(3)})
        try:
            red.ready({"x": torch.ones(3)})
            raise AssertionError("double hand-over accepted")
        except RuntimeError:
            pass
        try:
            red.finish({"y": torch.ones(3)})
            raise AssertionError("unknown gradient accepted")
        except RuntimeError:
            pass
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_reduction_world2():
    """shard.GradientReducer: buckets all-reduced asynchronously WHILE the backward runs (hand-over per block, last block
    first; biases at the end) give bit for bit the weights of the reduce-after-the-backward solver on both ranks, and match
    a single process on the whole batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {r: {m: tuple([torch.from_numpy(x) for x in part] for part in pair) for m, pair in d.items()} for r, d in got.items()}
    for kind in (0, 1):
        for a, b, c, d in zip(got[0]["overlapped"][kind], got[1]["overlapped"][kind], got[0]["after"][kind],
                              got[1]["after"][kind]):
            assert torch.equal(a, b) and torch.equal(c, d)
            assert torch.allclose(a, c, rtol=1e-6, atol=1e-8)
    # single process, whole batch
    from text_to_sound_synthesis_amd.modeling.solver import GradClipWindow, Solver
    g = torch.Generator().manual_seed(33)
    batches = [torch.randn(8, 6, generator=g) for _ in range(3)]
    ws, bs = _block_model()
    s = Solver(_BlockStep(ws, bs), lr=1e-2, clip_grad_norm=GradClipWindow(max_norm=0.5), reducer=shard.GradientReducer())
    for b in batches:
        s.step(b)                          # no process group: the reducer is a no-op
    for a, w in zip(got[0]["overlapped"][0], ws):
        assert torch.allclose(a, w, rtol=1e-5, atol=1e-7)


# ---- bench.py's own launcher (round 5): `python bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself -----------
def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_self_launcher_starts_two_gloo_ranks():
    """The launcher path end to end on CPU: bench.py --gpus 2 (no WORLD_SIZE) re-executes itself under
    torch.distributed.run with 2 ranks; rank 0 prints one line whose `rccl_ranks` is what an all-reduce of ones saw, after
    the path's scatter / gather / gradient-bucket collectives ran on stand-in tensors (gloo here, nccl on GPUs)."""
    import json
    r = _run_bench(["--gpus", "2", "--collectives-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                         # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 2 and out["n_gpus"] == 2 and out["backend"] == "gloo"
    assert out["gather_bytes"] == 16 * 217088 * 4 and out["scatter_ms"] > 0 and out["grad_bucket_ms"] > 0


def test_bench_refuses_more_ranks_than_devices():
    """No silent one-GPU run: asking for 2 ranks where fewer devices are visible fails loudly with exit code 2."""
    r = _run_bench(["--gpus", "2"])
    assert r.returncode == 2 and "2 ranks requested" in r.stderr and not r.stdout.strip()


def test_bench_refuses_world_size_that_contradicts_gpus():
    r = _run_bench(["--gpus", "2", "--collectives-selftest"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


# ---- after the timed region: side legs may fail or hang, the line is printed regardless (round 5) ------------------------
def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_side_legs_file_their_errors_and_stop_entering_collectives(capsys):
    """bench.run_side_legs: a failing leg becomes line[name] = {"error": ...}; one GPU: the remaining legs still run; N ranks:
    after a failed COLLECTIVE leg this rank enters no further collective (the others are still inside the failed one) but
    still runs its local legs, and the caller is told (False) not to tear the process group down."""
    bench = _bench_module()
    ran = []

    def ok(name):
        return lambda: ran.append(name)

    def boom():
        raise RuntimeError("rank 1 went away")
    legs = [("host_copy", ok("host_copy"), True, False), ("roofline", boom, True, False), ("stage_ms", ok("stage_ms"), True, True),
            ("rccl", boom, True, True), ("train", ok("train"), True, True), ("skipped", ok("skipped"), False, False),
            ("cpu_baseline", ok("cpu_baseline"), True, False)]
    line = {"value": 1.0}
    assert bench.run_side_legs(legs, 1, 0, line) is True              # one GPU: every leg is attempted
    assert ran == ["host_copy", "stage_ms", "train", "cpu_baseline"]
    assert line["roofline"]["error"].startswith("RuntimeError") and "rccl" in line and line["value"] == 1.0
    ran.clear()
    line = {"value": 1.0}
    assert bench.run_side_legs(legs, 2, 0, line) is False             # N ranks: nothing collective after the failed one
    assert ran == ["host_copy", "stage_ms", "cpu_baseline"] and "error" in line["rccl"] and "train" not in line
    line = {}
    assert bench.run_side_legs(legs[:3], 2, 1, line) is True and line == {}   # a local leg's failure on rank 1: filed on stderr only
    assert "side leg 'roofline' failed on rank 1" in capsys.readouterr().err


def test_headline_guard_prints_the_line_when_a_side_leg_hangs():
    """bench.HeadlineGuard: the normal path (finish() -> True, timer cancelled, nothing emitted), and the hang: the timer emits
    the line with the reason and ends the process with exit code 0 -- run for real in a child process whose "side leg" sleeps."""
    import json
    import subprocess
    import sys
    import time
    bench = _bench_module()
    got = []
    g = bench.HeadlineGuard(30.0, got.append, exit_fn=got.append).start()
    assert g.finish() is True and not got and g.finish() is False
    g = bench.HeadlineGuard(0.05, got.append, exit_fn=got.append).start()
    time.sleep(0.5)
    assert got == ["side legs unfinished after 0 s", 0] and g.finish() is False
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = ("import importlib.util, json, sys, time\n"
             "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
             "line = {'metric': 'clips/s', 'value': 23.0}\n"
             "g = b.HeadlineGuard(0.3, lambda why: print(json.dumps(dict(line, incomplete=why)), flush=True)).start()\n"
             "time.sleep(30)\n"
             "print('never reached')\n") % os.path.join(root, "bench.py")
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "never reached" not in r.stdout
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] == 23.0 and out["incomplete"].startswith("side legs unfinished")
