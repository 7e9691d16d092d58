"""Config plumbing: the reference builds every module from `{target: dotted.path, params: {...}}`
dicts (sound_synthesis/utils/misc.py:125-132).  The same YAML files work here: targets that name a
reference class on the generation path are redirected to the HIP-backed drop-in of the same name."""
import importlib

_PKG = __name__.rsplit(".", 1)[0]

# reference dotted path -> module in this package (class name is kept)
_ALIASES = {
    "sound_synthesis.modeling.models.dalle_spec.DALLE": _PKG + ".modeling.dalle",
    "sound_synthesis.modeling.transformers.diffusion_transformer.DiffusionTransformer": _PKG + ".modeling.diffusion",
    "sound_synthesis.modeling.transformers.transformer_utils.Text2ImageTransformer": _PKG + ".modeling.transformer",
    "sound_synthesis.modeling.embeddings.dalle_mask_image_embedding.DalleMaskImageEmbedding": _PKG + ".modeling.transformer",
    "sound_synthesis.modeling.codecs.spec_codec.vqgan.VQModel": _PKG + ".modeling.vqgan",
    "specvqgan.modules.transformer.permuter.ColumnMajor": _PKG + ".modeling.vqgan",
    "vocoder.modules.Generator": _PKG + ".modeling.vocoder",
    "sound_synthesis.modeling.codecs.text_codec.tokenize.Tokenize": _PKG + ".tokenizer",
    "sound_synthesis.modeling.embeddings.clip_text_embedding.CLIPTextEmbedding": _PKG + ".modeling.clip_text",
}
# parts of the reference config that are not on this path: built as None
_DEFERRED = (
    "specvqgan.modules.losses.DummyLoss",
)


def instantiate_from_config(config):
    if config is None:
        return None
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    target = config["target"]
    if target in _DEFERRED:
        return None
    module, cls = target.rsplit(".", 1)
    module = _ALIASES.get(target, module)
    return getattr(importlib.import_module(module), cls)(**config.get("params", dict()))


def load_yaml_config(path):
    import yaml
    with open(path) as f:
        return yaml.full_load(f)


def build_model(config, args=None):
    """sound_synthesis/modeling/build.py:4-5"""
    return instantiate_from_config(config["model"])


def default_config(n_layer=19, diffusion_step=100, n_embed=256, with_clip=False, bpe_path=None):
    """The shapes of Diffsound/evaluation/caps_text.yaml (values cited in SURVEY.md section 8),
    expressed with this package's own class paths.  with_clip attaches the text stage (BPE tokenizer +
    CLIP ViT-B/32 text tower, caps_text.yaml:30-41,67-76); without it the caption conditioning is
    passed in as `condition_embed_token`.  bpe_path: the tokenizer's merge table (default: DIFFSOUND_BPE_PATH /
    CLIP's full table; tokenizer.CLOSED_VOCAB_PATH = the closed vocabulary of the synthetic captions)."""
    m = _PKG + ".modeling."
    cfg = _default_config(m, n_layer, diffusion_step, n_embed)
    if with_clip:
        p = cfg["model"]["params"]
        p["condition_codec_config"] = {"target": _PKG + ".tokenizer.Tokenize", "params": {
            "context_length": 77, "add_start_and_end": True, "with_mask": True, "pad_value": 0,
            "clip_embedding": False, "tokenizer_config": {"params": {"end_idx": 49152}}, "bpe_path": bpe_path}}
        p["diffusion_config"]["params"]["condition_emb_config"] = {
            "target": m + "clip_text.CLIPTextEmbedding", "params": {
                "clip_name": "ViT-B/32", "num_embed": 49408, "normalize": True, "pick_last_embedding": False,
                "keep_seq_len_dim": False, "additional_last_embedding": False, "embed_dim": 512}}
    return cfg


def _default_config(m, n_layer, diffusion_step, n_embed):
    return {"model": {"target": m + "dalle.DALLE", "params": {
        "content_info": {"key": "image"},
        "condition_info": {"key": "text"},
        "content_codec_config": {"target": m + "vqgan.VQModel", "params": {
            "embed_dim": 256, "n_embed": n_embed, "ckpt_path": None,
            "ddconfig": {"double_z": False, "z_channels": 256, "resolution": 848, "in_channels": 1,
                         "out_ch": 1, "ch": 128, "ch_mult": [1, 1, 2, 2, 4], "num_res_blocks": 2,
                         "attn_resolutions": [53], "dropout": 0.0},
            "lossconfig": None}},
        "first_stage_permuter_config": {"target": m + "vqgan.ColumnMajor", "params": {"H": 5, "W": 53}},
        "condition_codec_config": None,
        "diffusion_config": {"target": m + "diffusion.DiffusionTransformer", "params": {
            "diffusion_step": diffusion_step, "alpha_init_type": "alpha1",
            "auxiliary_loss_weight": 5.0e-4, "adaptive_auxiliary_loss": True, "mask_weight": [1, 1],
            "transformer_config": {"target": m + "transformer.Text2ImageTransformer", "params": {
                "attn_type": "selfcross", "n_layer": n_layer, "condition_seq_len": 77,
                "content_seq_len": 265, "content_spatial_size": [5, 53], "n_embd": 1024,
                "condition_dim": 512, "n_head": 16, "attn_pdrop": 0.0, "resid_pdrop": 0.0,
                "block_activate": "GELU2", "timestep_type": "adalayernorm", "mlp_hidden_times": 4}},
            "condition_emb_config": None,
            "content_emb_config": {"target": m + "transformer.DalleMaskImageEmbedding", "params": {
                "num_embed": n_embed, "spatial_size": [5, 53], "embed_dim": 1024,
                "trainable": True, "pos_emb_type": "embedding"}}}}}}}
