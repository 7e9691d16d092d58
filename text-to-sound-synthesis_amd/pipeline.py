"""Drop-in for Diffsound/evaluation/generate_samples_batch.py:class Diffsound (:42-187).

Same constructor (config, path, ckpt_vocoder) and checkpoint conventions: ckpt['model'] ->
DALLE.load_state_dict(strict=False), ckpt['ema'] overlaid on model.get_ema_model() (:69-85); the
vocoder is Generator(n_mel_channels, ngf, n_residual_layers) as `<ckpt_vocoder>/args.yml` says + best_netG.pt
(:29-40), and NO vocoder (.npy only, no .wav) when ckpt_vocoder is falsy (:53-56).  `generate_sample_with_condition` is the
tensor-returning form of the reference's file-writing drivers: mel and waveform stay on the GPU and
the vocoder runs on the whole batch.
"""
import os

import torch

from .config import build_model, default_config, load_yaml_config
from .modeling.vocoder import Generator


def read_vocoder_args(path):
    """(n_mel_channels, ngf, n_residual_layers) from a MelGAN `args.yml` (generate_samples_batch.py:34-36).  The reference
    unpickles the training script's argparse.Namespace with yaml.UnsafeLoader; this reads only the three integer fields it
    uses, as plain `key: int` lines (the `!!python/object:argparse.Namespace` tag line is skipped, nothing is executed).
    A missing field keeps the reference configuration's value (80, 32, 3)."""
    import re
    vals = {"n_mel_channels": 80, "ngf": 32, "n_residual_layers": 3}
    with open(path, "r") as f:
        for line in f:
            m = re.match(r"^\s*(n_mel_channels|ngf|n_residual_layers)\s*:\s*(\d+)\s*(#.*)?$", line)
            if m:
                vals[m.group(1)] = int(m.group(2))
    return vals["n_mel_channels"], vals["ngf"], vals["n_residual_layers"]


def load_vocoder(ckpt_vocoder, eval_mode=True):
    """generate_samples_batch.py:29-40: `<ckpt_vocoder>/best_netG.pt` into the Generator `<ckpt_vocoder>/args.yml`
    describes (no args.yml: the reference configuration 80 / 32 / 3).  A wrong directory raises."""
    d = str(ckpt_vocoder)
    f = os.path.join(d, "best_netG.pt")
    if not os.path.exists(f):
        raise FileNotFoundError("vocoder checkpoint not found: %s" % f)
    a = os.path.join(d, "args.yml")
    g = Generator(*(read_vocoder_args(a) if os.path.exists(a) else (80, 32, 3)))
    g.load_state_dict(torch.load(f, map_location="cpu", weights_only=False))
    return {"model": g.eval() if eval_mode else g}


class Diffsound:
    def __init__(self, config=None, path=None, ckpt_vocoder=None, device="cuda", random_vocoder=False):
        """ckpt_vocoder falsy: `self.vocoder = None` and the drivers write `.npy` only, as the reference does (:53-56).
        random_vocoder=True (tests / benchmarks without a checkpoint) builds a random-weight Generator(80, 32, 3) instead."""
        cfg = default_config(with_clip=True) if config is None else (load_yaml_config(config) if isinstance(config, str) else config)
        self.model = build_model(cfg)
        self.epoch = 0
        if path is not None:       # path=None: seeded / random weights on purpose (tests, bench); a wrong path raises
            if not os.path.exists(path):
                raise FileNotFoundError("Diffsound checkpoint not found: %s" % path)
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
            self.epoch = ckpt.get("last_epoch", ckpt.get("epoch", 0))
            self.model.load_state_dict(ckpt["model"], strict=False)
            if "ema" in ckpt:
                self.model.get_ema_model().load_state_dict(ckpt["ema"], strict=False)
        self.model = self.model.to(device).eval()
        for p in self.model.parameters():
            p.requires_grad = False
        if ckpt_vocoder:
            self.vocoder = load_vocoder(ckpt_vocoder)["model"].to(device)
        elif random_vocoder:
            self.vocoder = Generator(80, 32, 3).eval().to(device)
        else:
            self.vocoder = None
        if self.vocoder is not None:
            for p in self.vocoder.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def generate_sample_with_condition(self, cond, truncation_rate=0.85, replicate=1, fast=False, caption_ids=None,
                                       seed=None):
        """Captions -> (mel01 f32[B,80,848], wave f32[B,1,217088] -- None without a vocoder --, tokens), everything left on the GPU.
        `cond` is a list of caption strings (needs the text stage: tokenizer + CLIP in the config),
        token ids i64[B,77], or caption embeddings f32[B,77,512].  fast=n selects the skip-step sampler with
        skip_step n-1, spelled like the reference's drivers (generate_samples_batch.py:100-103,148-151).
        caption_ids (one global index per caption) switches the sampler to per-caption in-kernel noise: a caption's clip
        then does not depend on the batch it is generated in (DiffusionTransformer.rng_mode); replicate r of caption i
        draws as caption id ids[i] + r * 2^24."""
        if isinstance(cond, (list, tuple, str)):
            batch = {"text": [cond] if isinstance(cond, str) else list(cond)}
        elif cond.dtype == torch.long:
            batch = {"condition_token": cond}
        else:
            batch = {"condition_embed_token": cond}
        if caption_ids is not None:
            batch["caption_ids"] = caption_ids
        if seed is not None:
            batch["seed"] = seed
        out = self.model.generate_content(batch=batch, filter_ratio=0, replicate=replicate, content_ratio=1,
                                          return_att_weight=False,
                                          sample_type="top" + str(truncation_rate) + ("r,fast" + str(fast - 1) if fast else "r"))
        mel = out["content"]                                   # [B,1,80,848] in ~[-1,1]
        wave = None if self.vocoder is None else self.vocoder(mel[:, 0], scale=0.5, shift=0.5)   # spec = (x+1)/2, :182
        return (mel[:, 0] + 1) / 2, wave, out["content_token"]

    @torch.no_grad()
    def inference_generate_sample_with_condition(self, text, truncation_rate, save_root, batch_size, fast=False):
        """The reference's single-caption driver, same signature and behaviour (generate_samples_batch.py:89-123):
        ONE caption `text`, sampled `replicate = 10` times (hard-coded there, :111; `batch_size` is accepted and
        unused, as in the reference), results written under `save_root/str(text)/` as `000000`, `000001`, ...
        The reference writes `.png` through an image-era uint8 cast that cannot represent a [-1,1] spectrogram
        (`Image.fromarray` rejects the [80,848,1] array); this drop-in writes what the batch driver writes instead:
        `{n:06d}.npy` (mel in [0,1], f32[80,848]) and, with a vocoder, `{n:06d}.wav` (22 050 Hz PCM_24).  Returns the paths."""
        import numpy as np
        os.makedirs(save_root, exist_ok=True)
        save_root_ = os.path.join(save_root, str(text))
        os.makedirs(save_root_, exist_ok=True)
        mel01, wave, _ = self.generate_sample_with_condition([text], truncation_rate, replicate=10, fast=fast)
        mel01, wave = mel01.cpu().numpy(), None if wave is None else wave[:, 0].cpu().numpy()
        written = []
        for b in range(mel01.shape[0]):
            path = os.path.join(save_root_, str(b).zfill(6))
            np.save(path + ".npy", mel01[b])
            if wave is not None:
                write_wav_pcm24(path + ".wav", wave[b], 22050)
            written.append(path)
        return written

    @staticmethod
    def read_tsv(val_path):
        """file_name,caption CSV -> {file_name: [captions]} (generate_samples_batch.py:125-141)."""
        import csv
        caps = {}
        with open(val_path, newline="") as f:
            for row in csv.DictReader(f):
                caps.setdefault(row["file_name"], []).append(row["caption"])
        return caps

    @torch.no_grad()
    def generate_sample(self, val_path, truncation_rate, save_root, fast=False, replicate=2):
        """The reference's file-writing driver (generate_samples_batch.py:143-187): per audio file, all of
        its captions x `replicate` are sampled in one batch; every sample is written as
        `{base}_mel_sample_{i}.npy` (mel in [0,1], f32[80,848]) and -- if there is a vocoder (:183) --
        `{base}_mel_sample_{i}.wav` (22 050 Hz, PCM_24).  Unlike the reference the vocoder runs on the whole batch at once."""
        import numpy as np
        os.makedirs(save_root, exist_ok=True)
        written = []
        n_seen = 0                       # running caption index over the whole table = the global caption id
        philox = self.model.transformer.rng_mode == "philox"
        for key, captions in self.read_tsv(val_path).items():
            base = key.split(".")[0] + "_mel_sample_"
            ids = list(range(n_seen, n_seen + len(captions))) if philox else None
            n_seen += len(captions)
            mel01, wave, _ = self.generate_sample_with_condition(list(captions), truncation_rate, replicate, fast=fast,
                                                                 caption_ids=ids)
            mel01, wave = mel01.cpu().numpy(), None if wave is None else wave[:, 0].cpu().numpy()
            for i in range(mel01.shape[0]):
                path = os.path.join(save_root, base + str(i))
                np.save(path + ".npy", mel01[i])
                if wave is not None:
                    write_wav_pcm24(path + ".wav", wave[i], 22050)
                written.append(path)
        return written


def write_wav_pcm24(path, samples, rate):
    """Mono float waveform in [-1, 1] -> RIFF/WAVE with 24-bit little-endian PCM, as
    soundfile.write(path, x, rate, 'PCM_24') produces (generate_samples_batch.py:186)."""
    import struct

    import numpy as np
    x = np.asarray(samples, dtype=np.float64).reshape(-1)
    q = np.clip(np.rint(x * 8388608.0), -8388608, 8388607).astype("<i4")   # libsndfile: scale 2^23, clip
    raw = q.view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 3, 3, 24))
        f.write(b"data" + struct.pack("<I", len(raw)))
        f.write(raw)
