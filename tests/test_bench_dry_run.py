"""bench.py's main() end to end WITHOUT a GPU: the control flow after the timed region (side-leg table, error filing, the
watchdog's normal path, the one JSON line) with the three device stages replaced by stand-ins of the right shapes.  The
device work itself is what the `-m gpu` tests and the GPU runs under profiles/ cover; this test exists because the side legs
were restructured after the round's GPU budget was spent."""
import importlib.util
import json
import os
import sys
import types

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
