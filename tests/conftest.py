import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# measured parity figures (N1 flips / identical clips / mel / RMS, text-stage agreement ...): tests append one line each
# and the terminal summary prints them, so a driver log that only keeps pytest's tail still carries the numbers
PARITY_SUMMARY = []


def parity_line(text):
    PARITY_SUMMARY.append(text)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if PARITY_SUMMARY:
        terminalreporter.section("measured parity (vs reference-generated goldens)")
        for line in PARITY_SUMMARY:
            terminalreporter.write_line(line)


@pytest.fixture(autouse=True)
def _grad_mode(request):
    """Grad mode is per test, never process-global: modules that set NO_GRAD = True run their tests under
    torch.no_grad(); every other test gets grad enabled, whatever ran before it in the same pytest process."""
    with torch.set_grad_enabled(not getattr(request.module, "NO_GRAD", False)):
        yield


def golden(name):
    return {k: torch.from_numpy(v) if v.ndim else v
            for k, v in np.load(os.path.join(GOLDEN, name + ".npz")).items()}


def key_contract():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        c = json.load(f)
    for extra in ("state_dict_keys_encoder.json",      # scope row 8f-2, generated separately
                  "state_dict_keys_k512.json"):         # the 512-entry codebook build (BASELINE configs[3])
        with open(os.path.join(GOLDEN, extra)) as f:
            c.update(json.load(f))
    return c


_SD_CACHE = {}


def synth_sd(which, n_layer=19, seed=0, profile="init"):
    """Synthetic weights for the reference's parameter names (trimmed to n_layer blocks).  profile="trained": the
    denoiser's tensors get trained-like statistics (text_to_sound_synthesis_amd/synth.py)."""
    from text_to_sound_synthesis_amd.synth import synth_state_dict
    ck = (which, n_layer, seed, profile)
    if ck not in _SD_CACHE:
        shapes = key_contract()[which]["params"]
        if which == "dalle_k512" and n_layer > 2:     # the contract file lists the 2-layer build: blocks are K-independent
            shapes = dict(key_contract()["dalle"]["params"], **shapes)
        if which in ("dalle", "dalle_k512"):
            keep = {}
            for k, s in shapes.items():
                if ".blocks." in k and int(k.split(".blocks.")[1].split(".")[0]) >= n_layer:
                    continue
                keep[k] = s
            shapes = keep
        _SD_CACHE[ck] = synth_state_dict(shapes, seed, profile)
    return _SD_CACHE[ck]


@pytest.fixture(scope="session")
def sd_dalle_l2():
    return synth_sd("dalle", 2)


@pytest.fixture(scope="session")
def sd_encoder():
    """content_codec.encoder.* + content_codec.quant_conv.* (the VQ encode side)"""
    return synth_sd("encoder")


@pytest.fixture(scope="session")
def sd_dalle_l19():
    return synth_sd("dalle", 19)


@pytest.fixture(scope="session")
def sd_vocoder():
    return synth_sd("generator")
