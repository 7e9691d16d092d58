"""TEST INFRASTRUCTURE -- not shipped, never imported by the product path.

Harness that imports the *unmodified* reference from /root/reference/Diffsound on
CPU (SURVEY.md §8c recipe).  It exists only in the build container: it validates
oracle/diffsound_oracle.py and generates tests/golden/*.npz (oracle/make_golden.py).
Nothing on the GPU box may import this file (/root/reference is absent there).

Stubs (modules the reference imports but the hot path never uses):
  pytorch_lightning  -- spec_codec/vqgan.py:3,11 (LightningModule = nn.Module)
  librosa(.filters)  -- vocoder/modules.py:4 (Audio2Mel only)
  ftfy               -- clip/simple_tokenizer.py:6
  torchvision(.transforms), PIL is real
Patches:
  torch.Tensor.cuda -> identity (transformer_utils.py:434 hard-codes t.cuda())
"""
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("DIFFSOUND_REF", "/root/reference/Diffsound")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "sound_synthesis"))


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    if "pytorch_lightning" not in sys.modules:
        stub("pytorch_lightning", LightningModule=torch.nn.Module)
    if "librosa" not in sys.modules:
        lf = stub("librosa.filters", mel=lambda *a, **k: None)
        stub("librosa", filters=lf)
    if "ftfy" not in sys.modules:
        stub("ftfy", fix_text=lambda s: s)
    if "torchvision" not in sys.modules:
        class _T:  # placeholders for `from torchvision.transforms import ...`
            def __init__(self, *a, **k):
                pass
        tr = stub("torchvision.transforms", Compose=_T, Resize=_T, CenterCrop=_T,
                  ToTensor=_T, Normalize=_T)
        stub("torchvision", transforms=tr)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


def ref_config(n_layer=19, diffusion_step=100, n_embed=256, with_clip=False):
    """The reference's own evaluation/caps_text.yaml, with the absent checkpoint
    path removed and -- unless with_clip -- CLIP replaced by 'condition_embed' injection.
    with_clip keeps the file's condition_emb_config (CLIPTextEmbedding inside the
    DiffusionTransformer, caps_text.yaml:67-76): the training batch's own route."""
    import yaml
    with open(os.path.join(REF_ROOT, "evaluation", "caps_text.yaml")) as f:
        cfg = yaml.full_load(f)
    p = cfg["model"]["params"]
    p["content_codec_config"]["params"]["ckpt_path"] = None
    p["content_codec_config"]["params"]["n_embed"] = n_embed
    d = p["diffusion_config"]["params"]
    d["diffusion_step"] = diffusion_step
    d["transformer_config"]["params"]["n_layer"] = n_layer
    d["content_emb_config"]["params"]["num_embed"] = n_embed
    if not with_clip:
        d["condition_emb_config"] = None
    return cfg


def build_dalle(n_layer=19, diffusion_step=100, n_embed=256, seed=0, with_encoder=False, with_clip=False):
    """Reference DALLE carrying synth weights keyed by state-dict name.  with_encoder also gives the VQ encoder +
    quant_conv synth weights (scope row 8f-2); with_clip builds the reference's CLIPTextEmbedding inside the
    DiffusionTransformer (random-init ViT-B/32 text tower, see build_clip_text; its keys are
    transformer.condition_emb.*, the same synth values build_clip_text pours).  Weights are keyed by name, so nothing
    else changes."""
    install()
    from sound_synthesis.modeling.build import build_model
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from text_to_sound_synthesis_amd.synth import synth_init_
    cfg = ref_config(n_layer, diffusion_step, n_embed, with_clip=with_clip)
    if with_clip:
        _redirect_clip_load()
    model = build_model(cfg).eval()
    # the VQ encoder / loss are not on the path; leave them at their defaults
    synth_init_(model, seed=seed, skip=("content_codec.loss.",) if with_encoder else
                ("content_codec.encoder.", "content_codec.quant_conv.", "content_codec.loss."))
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def build_vocoder(seed=0):
    install()
    from vocoder.modules import Generator
    from text_to_sound_synthesis_amd.synth import synth_init_
    g = Generator(80, 32, 3).eval()
    synth_init_(g, seed=seed)
    for p in g.parameters():
        p.requires_grad_(False)
    return g


def _redirect_clip_load():
    from sound_synthesis.modeling.modules.clip import clip as clip_mod
    from sound_synthesis.modeling.modules.clip import model as clip_model
    clip_mod.load = lambda *a, **k: (clip_model.CLIP(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12), None)


def build_clip_text(seed=0):
    """The reference's CLIPTextEmbedding (fp16 ViT-B/32 text tower, per-token output) with synth weights.
    clip.load is redirected to a random-init CLIP of the ViT-B/32 shape: the real ViT-B-32.pt path is
    hard-coded to the authors' cluster (modules/clip/clip.py:96)."""
    install()
    from text_to_sound_synthesis_amd.synth import synth_init_
    _redirect_clip_load()
    from sound_synthesis.modeling.embeddings.clip_text_embedding import CLIPTextEmbedding
    m = CLIPTextEmbedding(clip_name="ViT-B/32", num_embed=49408, normalize=True, pick_last_embedding=False,
                          keep_seq_len_dim=False, additional_last_embedding=False, embed_dim=512).eval()
    synth_init_(m, seed=seed, prefix="transformer.condition_emb.")
    return m


def reference_tokenize(texts):
    install()
    from sound_synthesis.modeling.codecs.text_codec.tokenize import Tokenize
    t = Tokenize(context_length=77, add_start_and_end=True, with_mask=True, pad_value=0, clip_embedding=False,
                 tokenizer_config={"target": "sound_synthesis.modeling.modules.clip.simple_tokenizer.SimpleTokenizer",
                                   "params": {"end_idx": 49152}})
    return t.get_tokens(texts)
