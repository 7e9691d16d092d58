"""Drop-in for sound_synthesis/modeling/models/dalle_spec.py:DALLE (generation side).

generate_content() keeps the reference's keyword signature and `sample_type` mini-language
(:179-247): "top{r}r" installs top-r truncation -- natively, as a kernel argument, instead of the
reference's monkey-patched predict_start wrapper (:208-210).  The caption conditioning is either text
(`batch['text']`: BPE tokenizer + the HIP CLIP text tower, when the config builds them), already tokenised
captions (`condition_token` i64[B, 77]) or the embedding itself (`condition_embed_token` f32[B, 77, 512], what
CLIPTextEmbedding.forward returns), in `batch` or as `condition=`.
"""
import torch
from torch import nn

from ..config import instantiate_from_config


class DALLE(nn.Module):
    def __init__(self, *, content_info={"key": "image"}, condition_info={"key": "text"}, content_codec_config,
                 condition_codec_config, first_stage_permuter_config, diffusion_config):
        super().__init__()
        self.content_info = content_info
        self.condition_info = condition_info
        self.content_codec = instantiate_from_config(content_codec_config)
        self.condition_codec = instantiate_from_config(condition_codec_config)  # Tokenize: section 8f-1
        self.transformer = instantiate_from_config(diffusion_config)
        self.first_stage_permuter = instantiate_from_config(first_stage_permuter_config)
        self.truncation_forward = False

    @property
    def device(self):
        return self.transformer.device

    def get_ema_model(self):
        return self.transformer

    @torch.no_grad()
    def get_tokens(self, spec):
        """mel image [B, 1, 80, 848] -> (quant_z, token ids [B, 265] in sequence order) (dalle_spec.py:70-77)."""
        quant_z, _, info = self.content_codec.encode(spec)
        indices = self.first_stage_permuter(info[2].view(quant_z.shape[0], -1))
        self.zshape = quant_z.shape
        return quant_z, indices

    @torch.no_grad()
    def prepare_content(self, batch, with_mask=False):
        """batch[content key] -> {'content_token', 'content_quant'} (dalle_spec.py:107-126, with_mask=False branch)."""
        if with_mask:
            raise NotImplementedError("masked content encoding is not part of the sound pipeline (:118-120)")
        cont = batch[self.content_info["key"]]
        if torch.is_tensor(cont):
            cont = cont.to(self.device)
        quant_z, indices = self.get_tokens(cont)
        return {"content_token": indices, "content_quant": quant_z}

    @torch.no_grad()
    def prepare_input(self, batch):
        """condition + content of a training batch (dalle_spec.py:128-133)"""
        inp = self.prepare_condition(batch)
        inp.update(self.prepare_content(batch))
        return inp

    @torch.no_grad()
    def forward(self, batch, name="none", **kwargs):
        """batch {'image': mel f32[B,1,80,848], 'text' | 'condition_embed_token': ...} -> the transformer's
        {'logits', 'loss'} (dalle_spec.py:389-400).  Forward value only (no autograd through the HIP path)."""
        return self.transformer(self.prepare_input(batch), **kwargs)

    def decode_to_img(self, index, zshape, stage="first"):
        """tokens (sequence order) -> mel image [B, 1, 80, 848] (:80-91)."""
        assert stage == "first"
        return self.content_codec.decode_tokens(index, zshape[2], zshape[3])

    @torch.no_grad()
    def prepare_condition(self, batch, condition=None):
        cond = {}
        src = batch if condition is None else condition
        if torch.is_tensor(src):
            src = {"condition_embed_token": src}
        emb = src.get("condition_embed_token", src.get("embed_token"))
        tok = src.get("condition_token", src.get("token"))
        if emb is not None:
            cond["condition_embed_token"] = emb.to(self.device)
            cond["condition_token"] = None
        elif tok is not None:      # already tokenised captions i64[B, 77]
            cond["condition_token"] = tok.to(self.device)
        elif self.condition_codec is not None:
            for k, v in self.condition_codec.get_tokens(src[self.condition_info["key"]]).items():
                cond["condition_" + k] = v.to(self.device) if torch.is_tensor(v) else v
        else:
            raise NotImplementedError(
                "this model was built without a text codec (condition_codec_config = None): pass "
                "batch={'condition_embed_token': f32[B,77,512]} or {'condition_token': i64[B,77]}")
        return cond

    @torch.no_grad()
    def generate_content(self, *, batch, condition=None, filter_ratio=0.5, temperature=1.0, content_ratio=0.0,
                         replicate=1, return_att_weight=False, sample_type="top0.85r"):
        self.eval()
        condition = self.prepare_condition(batch=batch, condition=condition)
        if replicate != 1:
            for k in condition:
                if condition[k] is not None:
                    condition[k] = torch.cat([condition[k] for _ in range(replicate)], dim=0)
        parts = sample_type.split(",")
        tr = self.transformer
        if len(parts) > 1 and parts[1][:1] == "q":           # repeat-step sampler (:135-143, :205-206)
            tr.repeat_rate = float(parts[1].replace("q", ""))
        if parts[0][:3] == "top" and not self.truncation_forward:
            # installed once and sticky, like the reference's predict_start wrapper (:207-209)
            if parts[0][-1] == "p":
                tr.truncation_k, tr.truncation_r = int(parts[0][:-1].replace("top", "")), None
            elif parts[0][-1] == "r":
                tr.truncation_r, tr.truncation_k = float(parts[0][:-1].replace("top", "")), None
            else:
                print("wrong sample type")                    # the reference's reaction (:176-177)
            self.truncation_forward = True
        kw = dict(condition_token=condition.get("condition_token"), condition_mask=condition.get("condition_mask"),
                  condition_embed=condition.get("condition_embed_token"), content_token=None,
                  filter_ratio=filter_ratio, temperature=temperature, return_att_weight=return_att_weight,
                  return_logits=False, print_log=False, sample_type=sample_type)
        if batch.get("caption_ids") is not None:             # per-caption in-kernel noise (diffusion.py rng_mode)
            ids = torch.as_tensor(batch["caption_ids"], dtype=torch.long)
            kw["caption_ids"] = torch.cat([ids for _ in range(replicate)]) + \
                torch.arange(replicate).repeat_interleave(ids.numel()) * int(batch.get("caption_id_stride", 1 << 24))
        if batch.get("seed") is not None:
            kw["seed"] = int(batch["seed"])
        if len(parts) == 2 and parts[1][:4] == "fast":       # skip-step sampler (:211-222)
            trans_out = tr.sample_fast(skip_step=int(parts[1][4:]), **kw)
        else:
            trans_out = tr.sample(**kw)
        tokens = trans_out["content_token"]
        zshape = (tokens.shape[0], 256, 5, 53)   # hard-coded in the reference too (:236)
        content = self.decode_to_img(tokens, zshape)
        self.train()
        return {"content": content, "content_token": tokens}

    @torch.no_grad()
    def sample(self, batch, clip=None, temperature=1., return_rec=True, filter_ratio=[0, 0.5, 1.0], content_ratio=[1],
               return_att_weight=False, return_logits=False, sample_type="normal", **kwargs):
        """The trainer's logging sampler (dalle_spec.py:264-343): encode the batch's mel to tokens, optionally decode
        them back ('reconstruction_image'), and for every filter_ratio fr re-sample from those tokens diffused to
        t = int(T * fr) - 1 (fr = 0: from the all-[MASK] state) -> 'cond1_cont{cr}_fr{fr}_image'.  Only
        content_ratio = 1 is meaningful for the fixed 265-token grid (the reference slices the token sequence, which
        its own q_sample cannot take either)."""
        if return_att_weight:
            raise NotImplementedError("attention weights are never materialised on the HIP path")
        if sample_type == "debug":
            raise NotImplementedError("sample_debug is not part of the sound pipeline")
        self.eval()
        condition = self.prepare_condition(batch)
        content = self.prepare_content(batch)
        out = {"input_image": batch[self.content_info["key"]]}
        zshape = content["content_quant"].shape
        if return_rec:
            out["reconstruction_image"] = self.decode_to_img(content["content_token"], zshape)
        for fr in filter_ratio:
            for cr in content_ratio:
                n_tok = int(content["content_token"].shape[1] * cr)
                if n_tok < 0:
                    continue
                if n_tok != content["content_token"].shape[1] and int(self.transformer.num_timesteps * fr) > 0:
                    raise ValueError("content_ratio < 1 cannot be re-sampled: q_sample needs the whole token grid")
                trans_out = self.transformer.sample(
                    condition_token=condition.get("condition_token"), condition_mask=condition.get("condition_mask"),
                    condition_embed=condition.get("condition_embed_token"), content_token=content["content_token"][:, :n_tok],
                    filter_ratio=fr, temperature=temperature, return_att_weight=False, return_logits=return_logits,
                    content_logits=None, sample_type=sample_type, **kwargs)
                out["cond1_cont{}_fr{}_image".format(cr, fr)] = self.decode_to_img(trans_out["content_token"], zshape)
                if return_logits:
                    out["logits"] = trans_out["logits"]
        self.train()
        res = {"condition": batch.get(self.condition_info["key"])}
        res.update(out)
        return res

