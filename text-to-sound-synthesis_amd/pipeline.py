"""Drop-in for Diffsound/evaluation/generate_samples_batch.py:class Diffsound (:42-187).

Same constructor (config, path, ckpt_vocoder) and checkpoint conventions: ckpt['model'] ->
DALLE.load_state_dict(strict=False), ckpt['ema'] overlaid on model.get_ema_model() (:69-85); the
vocoder is Generator(80, 32, 3) + best_netG.pt (:29-40).  `generate_sample_with_condition` is the
tensor-returning form of the reference's file-writing drivers: mel and waveform stay on the GPU and
the vocoder runs on the whole batch.
"""
import os

import torch

from .config import build_model, default_config, load_yaml_config
from .modeling.vocoder import Generator


def load_vocoder(ckpt_vocoder, eval_mode=True):
    g = Generator(80, 32, 3)
    if ckpt_vocoder:
        sd = torch.load(os.path.join(str(ckpt_vocoder), "best_netG.pt"), map_location="cpu")
        g.load_state_dict(sd)
    return {"model": g.eval() if eval_mode else g}


class Diffsound:
    def __init__(self, config=None, path=None, ckpt_vocoder=None, device="cuda"):
        cfg = default_config() if config is None else (load_yaml_config(config) if isinstance(config, str) else config)
        self.model = build_model(cfg)
        self.epoch = 0
        if path and os.path.exists(path):
            ckpt = torch.load(path, map_location="cpu")
            self.epoch = ckpt.get("last_epoch", ckpt.get("epoch", 0))
            self.model.load_state_dict(ckpt["model"], strict=False)
            if "ema" in ckpt:
                self.model.get_ema_model().load_state_dict(ckpt["ema"], strict=False)
        self.model = self.model.to(device).eval()
        for p in self.model.parameters():
            p.requires_grad = False
        self.vocoder = load_vocoder(ckpt_vocoder)["model"].to(device)
        for p in self.vocoder.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def generate_sample_with_condition(self, cond, truncation_rate=0.85, replicate=1):
        """cond: f32[B,77,512] caption embeddings -> (mel01 f32[B,80,848], wave f32[B,1,217088], tokens)."""
        out = self.model.generate_content(batch={"condition_embed_token": cond}, filter_ratio=0,
                                          replicate=replicate, content_ratio=1, return_att_weight=False,
                                          sample_type="top" + str(truncation_rate) + "r")
        mel = out["content"]                                   # [B,1,80,848] in ~[-1,1]
        wave = self.vocoder(mel[:, 0], scale=0.5, shift=0.5)   # spec = (x+1)/2, :182
        return (mel[:, 0] + 1) / 2, wave, out["content_token"]

    inference_generate_sample_with_condition = generate_sample_with_condition
