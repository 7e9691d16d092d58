"""bench.py's main() end to end WITHOUT a GPU: the control flow after the timed region (side-leg table, error filing, the
watchdog's normal path, the one JSON line) with the three device stages replaced by stand-ins of the right shapes.  The
device work itself is what the `-m gpu` tests and the GPU runs under profiles/ cover; this test exists because the side legs
were restructured after the round's GPU budget was spent."""
import importlib.util
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeCuda:
    """torch.cuda as bench.py sees it on a one-GPU box, minus the GPU"""
    @staticmethod
    def is_available():
        return True

    @staticmethod
    def device_count():
        return 1

    @staticmethod
    def set_device(i):
        pass

    @staticmethod
    def synchronize(*a):
        pass

    @staticmethod
    def empty_cache():
        pass


class _TorchProxy(types.ModuleType):
    """`torch` for bench.py only: device("cuda", i) is the CPU, torch.cuda is _FakeCuda, everything else is torch"""

    def __init__(self):
        super().__init__("torch")

    def __getattr__(self, name):
        if name == "cuda":
            return _FakeCuda
        if name == "device":
            return lambda *a, **k: torch.device("cpu")
        return getattr(torch, name)


def test_bench_main_prints_one_line_when_every_device_leg_fails(monkeypatch, capsys):
    from text_to_sound_synthesis_amd.modeling import dalle, diffusion, vocoder
    B = 2
    monkeypatch.setattr(diffusion.DiffusionTransformer, "sample",
                        lambda self, **kw: {"content_token": torch.zeros(kw["condition_token"].shape[0], 265, dtype=torch.long)})
    monkeypatch.setattr(dalle.DALLE, "decode_to_img", lambda self, index, zshape, stage="first": torch.zeros(index.shape[0], 1, 80, 848))
    monkeypatch.setattr(vocoder.Generator, "forward", lambda self, mel, **kw: torch.zeros(mel.shape[0], 1, 217088))
    spec = importlib.util.spec_from_file_location("bench_dry", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.torch = _TorchProxy()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--batch", str(B), "--n-layer", "2", "--steps", "2", "--warmup", "1",
                                      "--no-cpu-baseline", "--side-leg-limit", "600"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    grad = torch.is_grad_enabled()
    try:
        bench.main()
    finally:
        torch.set_grad_enabled(grad)
    out, err = capsys.readouterr()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                                        # ONE JSON line
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in line
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and "incomplete" not in line
    assert line["config"]["global_batch"] == B
    # the stage split ran on the stand-ins; the legs that need the device failed, each under its own name, and cost nothing else
    assert set(line["stage_ms"]) >= {"scatter", "sample", "decode", "vocode", "gather"}
    for leg in ("host_copy", "roofline", "train"):
        assert "error" in line[leg], (leg, line[leg])
        assert "side leg %r failed on rank 0" % leg in err
        # ... for the reason expected on a box without a GPU -- not a NameError / TypeError of the leg's own code
        assert line[leg]["error"].startswith(("RuntimeError: No HIP GPUs", "DiffsoundHipError: tensor is not on a GPU")), line[leg]
    assert "cpu_baseline" not in line and "rccl" not in line


def _main_worker(rank, world, port, outdir, extra):
    """One rank of `bench.py --gpus 2` on CPU: gloo instead of RCCL, stand-in device stages, stdout / stderr into files."""
    import torch.distributed as dist
    from text_to_sound_synthesis_amd.modeling import dalle, diffusion, vocoder
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    so, se = open(os.path.join(outdir, "out%d" % rank), "w"), open(os.path.join(outdir, "err%d" % rank), "w")
    os.dup2(so.fileno(), 1)
    os.dup2(se.fileno(), 2)
    sys.stdout, sys.stderr = so, se
    diffusion.DiffusionTransformer.sample = \
        lambda self, **kw: {"content_token": torch.zeros(kw["condition_token"].shape[0], 265, dtype=torch.long)}
    dalle.DALLE.decode_to_img = lambda self, index, zshape, stage="first": torch.zeros(index.shape[0], 1, 80, 848)
    vocoder.Generator.forward = lambda self, mel, **kw: torch.full((mel.shape[0], 1, 217088), float(rank))
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend, **kw: real_init("gloo", rank=rank, world_size=world)     # no device_id on CPU
    spec = importlib.util.spec_from_file_location("bench_dry", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.torch = _TorchProxy()
    _FakeCuda.device_count = staticmethod(lambda: world)
    sys.argv = ["bench.py", "--gpus", str(world), "--batch", "2", "--n-layer", "2", "--steps", "2", "--warmup", "1",
                "--side-leg-limit", "600"] + list(extra)
    bench.main()           # with the training leg: ends in os._exit(0), the leg (collective) fails on every rank without GPUs
    sys.stdout.flush()
    sys.stderr.flush()


@pytest.mark.parametrize("train_leg", [True, False])
def test_bench_main_two_ranks_print_one_line_and_exit_0(tmp_path, train_leg):
    """The N-rank control flow of main() on CPU (gloo): timed loop with scatter / gather, the stage split's all-reduce, the
    communicator check, then (train_leg) a collective leg that fails on both ranks -- each rank skips what is collective after
    it, rank 0 prints the ONE line, both exit 0 without tearing the process group down -- or (--no-train-leg) the normal end:
    the line, then destroy_process_group."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    extra = [] if train_leg else ["--no-train-leg"]
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, str(tmp_path), extra)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, open(os.path.join(str(tmp_path), "err%d" % procs.index(p))).read()[-2000:]
    out0, out1 = (open(os.path.join(str(tmp_path), "out%d" % r)).read() for r in range(2))
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in out1.splitlines() if l.startswith("{")]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["value"] > 0 and "incomplete" not in line
    assert line["rccl"]["rccl_ranks"] == 2 and line["rccl"]["backend"] == "gloo" and "ms_per_step_per_rank" in line["rccl"]
    assert "max_over_ranks" in line["stage_ms"]
    assert line["roofline"]["error"].startswith("DiffsoundHipError")      # rank 0's local leg: costs nothing collective
    assert "host_copy" not in line and "cpu_baseline" not in line          # one-GPU legs
    err1 = open(os.path.join(str(tmp_path), "err1")).read()
    if train_leg:
        assert line["train"]["error"].startswith("RuntimeError: No HIP GPUs") and "side leg 'train' failed on rank 1" in err1
    else:
        assert "train" not in line and "side leg" not in err1
