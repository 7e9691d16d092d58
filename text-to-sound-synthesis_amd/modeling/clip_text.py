"""Drop-in for sound_synthesis/modeling/embeddings/clip_text_embedding.py:CLIPTextEmbedding (HIP).

The reference wraps the text tower of CLIP ViT-B/32 (modules/clip/model.py:236-354): token + position
embedding -> 12 x [x += MHA(ln_1(x), causal); x += c_proj(QuickGELU(c_fc(ln_2(x))))] -> ln_final ->
(pick_last_embedding False) per-token L2 normalisation; weights are cast to fp16 by
convert_weights (model.py:373-395) and activations follow.  State-dict keys are the reference's
(`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.N.{ln_1,ln_2,attn.in_proj_*,
attn.out_proj,mlp.c_fc,mlp.c_proj}.*`, `ln_final.*`, `text_projection`), so `ckpt['model']`'s
`transformer.condition_emb.*` entries load unchanged.

Numerics: storage is fp32 holding fp16-representable values; every op's output is rounded to the fp16
grid in the kernel epilogues (ds_*_f16, ds_gemm f16_round, ds_attention_ex), i.e. the reference's
fp16 semantics with fp32 accumulation.  The reference's own fp16 results differ between devices /
torch versions at the 1e-3 relative level, which is the parity tolerance used for this stage.
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import _lib


class _MHA(nn.Module):
    """Parameter container with nn.MultiheadAttention's names."""

    def __init__(self, d, heads):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        self.num_heads = heads


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = _MHA(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, 4 * d)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(4 * d, d))]))
        self.ln_2 = nn.LayerNorm(d)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class CLIPTextEmbedding(nn.Module):
    def __init__(self, clip_name="ViT-B/32", num_embed=49408, normalize=True, pick_last_embedding=True,
                 keep_seq_len_dim=False, additional_last_embedding=False, embed_dim=1024,
                 width=512, layers=12, heads=8, context_length=77):
        super().__init__()
        assert clip_name == "ViT-B/32"
        assert not pick_last_embedding and not additional_last_embedding and embed_dim == 512, \
            "Diffsound uses per-token embeddings (pick_last_embedding: False, embed_dim: 512)"
        self.num_embed, self.normalize = num_embed, normalize
        self.embed_dim = embed_dim
        self.width, self.heads, self.context_length = width, heads, context_length
        self.token_embedding = nn.Embedding(num_embed, width)
        self.positional_embedding = nn.Parameter(torch.randn(context_length, width) * 0.01)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.randn(width, width) * width ** -0.5)  # unused here
        self.trainable = False
        for p in self.parameters():
            p.requires_grad = False
        self._pk = None
        self._register_load_state_dict_pre_hook(lambda *a, **k: setattr(self, "_pk", None))

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def _packed(self):
        if self._pk is None:
            h = lambda t: t.detach().half().float().contiguous()     # convert_weights: fp16 weights
            f = lambda t: t.detach().float().contiguous()            # LayerNorm / embeddings stay fp32
            blocks = []
            for b in self.transformer.resblocks:
                blocks.append({"ln1": (f(b.ln_1.weight), f(b.ln_1.bias)),
                               "qkv": (h(b.attn.in_proj_weight), h(b.attn.in_proj_bias)),
                               "out": (h(b.attn.out_proj.weight), h(b.attn.out_proj.bias)),
                               "ln2": (f(b.ln_2.weight), f(b.ln_2.bias)),
                               "fc": (h(b.mlp.c_fc.weight), h(b.mlp.c_fc.bias)),
                               "proj": (h(b.mlp.c_proj.weight), h(b.mlp.c_proj.bias))})
            self._pk = {"tok": f(self.token_embedding.weight), "pos": f(self.positional_embedding),
                        "blocks": blocks, "lnf": (f(self.ln_final.weight), f(self.ln_final.bias))}
        return self._pk

    @torch.no_grad()
    def forward(self, index, **kwargs):
        """index i64[B, 77] -> f32[B, 77, 512] (rows L2-normalised), clip_text_embedding.py:65-88."""
        assert index.dim() == 2
        pk = self._packed()
        L = _lib.lib()
        B, T = index.shape
        D, M = self.width, B * T
        dev = pk["tok"].device
        idx = index.to(dev).clamp(min=0).contiguous()                # text[text < 0] = 0, :47
        x = torch.empty(M, D, device=dev)
        _lib.check(L.ds_embed_f16(_lib.ptr(idx), _lib.ptr(pk["tok"]), _lib.ptr(pk["pos"]), _lib.ptr(x), M, T, D,
                                  _lib.stream()))
        hn = torch.empty(M, D, device=dev)
        qkv = torch.empty(M, 3 * D, device=dev)
        att = torch.empty(M, D, device=dev)
        fc = torch.empty(M, 4 * D, device=dev)
        for blk in pk["blocks"]:
            _lib.check(L.ds_layernorm_f16(_lib.ptr(x), _lib.ptr(hn), M, D, _lib.ptr(blk["ln1"][0]),
                                          _lib.ptr(blk["ln1"][1]), _lib.stream()))
            _lib.gemm(hn, blk["qkv"][0], qkv, M, 3 * D, D, bias=blk["qkv"][1], f16_round=1)
            _lib.check(L.ds_attention_ex(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, 3 * D,
                                         qkv.data_ptr() + 8 * D, 3 * D, _lib.ptr(att), D, B, self.heads, T, T,
                                         float((D // self.heads) ** -0.5), 1, 1, _lib.stream()))
            _lib.gemm(att, blk["out"][0], x, M, D, D, bias=blk["out"][1], R=x, f16_round=1)
            _lib.check(L.ds_layernorm_f16(_lib.ptr(x), _lib.ptr(hn), M, D, _lib.ptr(blk["ln2"][0]),
                                          _lib.ptr(blk["ln2"][1]), _lib.stream()))
            _lib.gemm(hn, blk["fc"][0], fc, M, 4 * D, D, bias=blk["fc"][1], act=_lib.ACT_GELU2, f16_round=1)
            _lib.gemm(fc, blk["proj"][0], x, M, D, 4 * D, bias=blk["proj"][1], R=x, f16_round=1)
        _lib.check(L.ds_layernorm_f16(_lib.ptr(x), _lib.ptr(hn), M, D, _lib.ptr(pk["lnf"][0]), _lib.ptr(pk["lnf"][1]),
                                      _lib.stream()))
        if self.normalize:
            _lib.check(L.ds_l2norm_rows_f16(_lib.ptr(hn), _lib.ptr(x), M, D, _lib.stream()))
            hn = x
        return hn.view(B, T, D)
