"""Drop-in for the SpecVQGAN codec (decode side on the generation path; encode side = SURVEY.md 8f-2), HIP-backed.

Mirrors sound_synthesis/modeling/codecs/spec_codec/vqgan.py:VQModel (decode :62-65),
specvqgan/modules/diffusionmodules/model.py (Decoder :570-671, ResnetBlock :92-151, AttnBlock
:174-226, Upsample :37-52), specvqgan/modules/vqvae/quantize.py:VectorQuantizer.get_codebook_entry
(:88-103) and specvqgan/modules/transformer/permuter.py:ColumnMajor (:21-55) in attribute names and
state-dict keys.  nn.Conv2d / nn.GroupNorm members are parameter containers only.

Device layout: activations are channels-last [B, H, W, C] fp32.  Every conv is the gather-GEMM
(implicit GEMM, fp32 MFMA); GroupNorm is a statistics pass whose (scale, shift) the next conv applies
while staging its A tile together with swish; nearest-2x upsampling is index math inside the conv
loader; the 512-channel spatial attention runs as two batched GEMMs around a row softmax.
"""

import numpy as np
import torch
from torch import nn

from .. import _lib
from ..config import instantiate_from_config


HALO_MIN_ROWS = 5      # smallest image height the halo-tiled 3x3 kernel takes (the 5 x 53 token grid)


def Normalize(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, c, with_conv=True):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(c, c, 3, 1, 1)


class Downsample(nn.Module):
    """zero pad (0,1,0,1) + 3x3 stride-2 conv (diffusionmodules/model.py:60-77)"""

    def __init__(self, c, with_conv=True):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = nn.Conv2d(c, c, 3, 2, 0)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        assert not conv_shortcut and temb_channels == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.in_channels = c
        self.norm = Normalize(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, **ignore):
        super().__init__()
        assert out_ch == 1 and resamp_with_conv and not give_pre_end
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution = resolution
        block_in = ch * ch_mult[-1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, True)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)


class Encoder(nn.Module):
    """Parameter container with the reference's attribute names (diffusionmodules/model.py:410-465)."""

    def __init__(self, *, ch, out_ch=None, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignore):
        super().__init__()
        assert resamp_with_conv
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, True)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)


class VectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta=0.25):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    @torch.no_grad()
    def forward(self, z):
        """z f32[B, C, H, W] -> (z_q [B, C, H, W], loss, (perplexity, min_encodings [M, n_e], indices [M, 1])) as
        vqvae/quantize.py:31-86.  The code search runs on the GPU: z E^T by the gather-GEMM, then
        d = |z|^2 + |e|^2 - 2 z.e and the first argmin per row (ds_vq_argmin)."""
        B, Cc, H, W = z.shape
        E = self.embedding.weight.detach().float().contiguous()
        zf = z.permute(0, 2, 3, 1).contiguous().float().view(-1, Cc)
        M = zf.shape[0]
        ze = torch.empty(M, self.n_e, device=z.device)
        _lib.gemm(zf, E, ze, M, self.n_e, Cc)
        idx = torch.empty(M, device=z.device, dtype=torch.long)
        _lib.check(_lib.lib().ds_vq_argmin(_lib.ptr(zf), _lib.ptr(ze), _lib.ptr((E * E).sum(1).contiguous()),
                                           _lib.ptr(idx), None, M, Cc, self.n_e, _lib.stream()))
        z_q = E[idx]
        loss = torch.mean((z_q - zf) ** 2) * (1.0 + self.beta)              # forward value of :70
        min_encodings = torch.nn.functional.one_hot(idx, self.n_e).to(zf.dtype)
        e_mean = min_encodings.mean(0)
        perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
        z_q = z_q.view(B, H, W, Cc).permute(0, 3, 1, 2).contiguous()
        return z_q, loss, (perplexity, min_encodings, idx.unsqueeze(1))

    @torch.no_grad()
    def get_codebook_entry(self, indices, shape):
        """indices i64[B*H*W] in ROW-major cell order, shape (B, H, W, C) -> [B, C, H, W] (:88-103)."""
        B, H, W, Cc = shape
        # ds_codebook_gather reads the sequence in column-major order; present row-major cells that way
        seq = indices.view(B, H, W).transpose(1, 2).reshape(B, H * W).contiguous()
        out = torch.empty(B, H, W, Cc, device=indices.device, dtype=torch.float32)
        _lib.check(_lib.lib().ds_codebook_gather(_lib.ptr(seq), _lib.ptr(self.embedding.weight), _lib.ptr(out),
                                                 B, H, W, Cc, self.n_e, _lib.stream()))
        return out.permute(0, 3, 1, 2).contiguous()


class ColumnMajor(nn.Module):
    """Token order of spectrogram grids: sequence index w*H + h (permuter.py:21-55)."""

    def __init__(self, H, W):
        super().__init__()
        self.H, self.W = H, W
        idx = torch.tensor(np.arange(H * W).reshape(H, W).T.ravel())
        self.register_buffer("forward_shuffle_idx", idx)
        self.register_buffer("backward_shuffle_idx", torch.argsort(idx))

    def forward(self, x, reverse=False):
        return x[:, self.backward_shuffle_idx if reverse else self.forward_shuffle_idx]


def _with_split(w, b):
    """(w [N][K] fp32, bias) + the two fp16 planes of w * 2^s and 2^-s for the 3x3 convs' f16x2 kernel (+ the same planes
    in the fragment-packed layout of the halo-tiled kernel, where its shape constraints hold)"""
    w2, sc = _lib.split_f16x2(w)
    N, K = w.shape
    wq = _lib.pack_conv3x3_weights(w2, N, K // 9) if (K % 9 == 0 and N % 128 == 0 and (K // 9) % 32 == 0) else None
    return w, b, w2, sc, wq


def _pack_conv3(conv):
    w = conv.weight.detach().float()
    return _with_split(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous(), conv.bias.detach().float().contiguous())


def _pack_conv1(conv):
    w = conv.weight.detach().float()
    return w.reshape(w.shape[0], w.shape[1]).contiguous(), conv.bias.detach().float().contiguous()


class VQModel(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, n_embed=256, embed_dim=256, ckpt_path=None, ignore_keys=[],
                 image_key="image", colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quantize = VectorQuantizer(n_embed, embed_dim, beta=0.25)
        self.quant_conv = nn.Conv2d(ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self._pke = None
        self.ddconfig = dict(ddconfig)
        # arithmetic of the 3x3 convolutions: "f16x2" (default; fp32-class 3-pass fp16 split on the 16-bit matrix
        # cores, csrc/conv_f16x2.hip) or "fp32" (exact fp32 MFMA, csrc/gemm_f32.hip)
        self.conv_precision = "f16x2"
        # In "f16x2" mode every stride-1 3x3 conv (optionally behind a nearest-2x upsample) whose weights are fragment-packed
        # runs on the halo-tiled kernel (csrc/conv3x3_f16x2.hip: the input tile is activated and split once for all nine
        # taps, the GroupNorm statistics of the output come out of its epilogue); the shapes it does not take -- the
        # encoder's stride-2 convs, Cout = 1 (conv_out) -- and the whole "fp32" mode run on the tap-by-tap gather kernel
        # (csrc/conv_f16x2.hip / gemm_f32.hip).
        self._pk = None
        # samples decoded / encoded at once: bounds the full-resolution workspace (34.7 MB per sample and tensor,
        # ~15 GB live at 64) and the 32-bit element indices inside the kernels ([B][80][848][128] < 2^31 up to B=247);
        # measured at B=64: 0.287 / 0.247 / 0.224 / 0.208 s for chunks of 8 / 16 / 32 / 64 (tools/decode_ab.py)
        self.decode_chunk = 64
        self._register_load_state_dict_pre_hook(lambda *a, **k: (setattr(self, "_pk", None), setattr(self, "_pke", None)))
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.load_state_dict(sd, strict=False)  # loss / discriminator keys are not part of this module

    def _apply(self, fn, *a, **k):
        self._pk = self._pke = None
        return super()._apply(fn, *a, **k)

    # ---- packing -------------------------------------------------------------------------------------
    @torch.no_grad()
    def _packed(self):
        if self._pk is not None:
            return self._pk
        d = self.decoder
        pk = {"pq": _pack_conv1(self.post_quant_conv), "conv_in": _pack_conv3(d.conv_in)}

        def res(b):
            r = {"n1": (b.norm1.weight.detach().float().contiguous(), b.norm1.bias.detach().float().contiguous()),
                 "c1": _pack_conv3(b.conv1),
                 "n2": (b.norm2.weight.detach().float().contiguous(), b.norm2.bias.detach().float().contiguous()),
                 "c2": _pack_conv3(b.conv2), "cin": b.in_channels, "cout": b.out_channels}
            if b.in_channels != b.out_channels:
                r["nin"] = _pack_conv1(b.nin_shortcut)
            return r

        def att(a):
            qw, qb = _pack_conv1(a.q)
            kw, kb = _pack_conv1(a.k)
            return {"n": (a.norm.weight.detach().float().contiguous(), a.norm.bias.detach().float().contiguous()),
                    "qk": (torch.cat((qw, kw), 0).contiguous(), torch.cat((qb, kb), 0).contiguous()),
                    "v": _pack_conv1(a.v), "proj": _pack_conv1(a.proj_out), "c": a.in_channels}

        pk["mid"] = (res(d.mid.block_1), att(d.mid.attn_1), res(d.mid.block_2))
        pk["up"] = []
        for lvl in range(d.num_resolutions):
            u = d.up[lvl]
            pk["up"].append({"block": [res(b) for b in u.block], "attn": [att(a) for a in u.attn],
                             "upsample": _pack_conv3(u.upsample.conv) if lvl != 0 else None})
        pk["norm_out"] = (d.norm_out.weight.detach().float().contiguous(), d.norm_out.bias.detach().float().contiguous())
        w = d.conv_out.weight.detach().float()  # [1, C, 3, 3] -> [9 taps][C]
        pk["conv_out"] = (w[0].permute(1, 2, 0).reshape(9, -1).contiguous(), float(d.conv_out.bias.item()))
        self._pk = pk
        return pk

    @torch.no_grad()
    def _packed_encoder(self):
        if self._pke is not None:
            return self._pke
        gb = lambda n: (n.weight.detach().float().contiguous(), n.bias.detach().float().contiguous())

        def res(b):
            r = {"n1": gb(b.norm1), "c1": _pack_conv3(b.conv1), "n2": gb(b.norm2), "c2": _pack_conv3(b.conv2),
                 "cin": b.in_channels, "cout": b.out_channels}
            if b.in_channels != b.out_channels:
                r["nin"] = _pack_conv1(b.nin_shortcut)
            return r

        def att(a):
            qw, qb = _pack_conv1(a.q)
            kw, kb = _pack_conv1(a.k)
            return {"n": gb(a.norm), "qk": (torch.cat((qw, kw), 0).contiguous(), torch.cat((qb, kb), 0).contiguous()),
                    "v": _pack_conv1(a.v), "proj": _pack_conv1(a.proj_out), "c": a.in_channels}
        e = self.encoder
        # conv_in has ONE input channel: a direct 9-tap kernel (ds_conv3x3_c1), not an implicit GEMM with a contraction of 9
        # zero-padded to the conv kernel's 32-wide channel granule (rounds 2-5: 1.15 ms per 20 mels + a 174 MB zero fill)
        w = e.conv_in.weight.detach().float()                                   # [ch, 1, 3, 3]
        assert w.shape[1] == 1, "the SpecVQGAN encoder reads a one-channel mel"
        pk = {"conv_in": (w.reshape(w.shape[0], 9).contiguous(), e.conv_in.bias.detach().float().contiguous()),
              "down": [], "mid": (res(e.mid.block_1), att(e.mid.attn_1), res(e.mid.block_2)),
              "norm_out": gb(e.norm_out), "conv_out": _pack_conv3(e.conv_out), "quant": _pack_conv1(self.quant_conv)}
        for lvl in range(e.num_resolutions):
            dn = e.down[lvl]
            pk["down"].append({"block": [res(b) for b in dn.block], "attn": [att(a) for a in dn.attn],
                               "downsample": _pack_conv3(dn.downsample.conv) if lvl != e.num_resolutions - 1 else None})
        self._pke = pk
        return pk

    # ---- HIP op helpers (channels-last) ------------------------------------------------------------------
    @staticmethod
    def _gn(x, B, P, Cc, gamma_beta):
        dev = x.device
        sc = torch.empty(B, Cc, device=dev)
        sh = torch.empty(B, Cc, device=dev)
        part = getattr(x, "_gn_part", None)
        if part is not None:      # the conv that produced x left the partial sums of its output tiles: no statistics pass
            _lib.check(_lib.lib().ds_groupnorm_finish(_lib.ptr(part), B, part.shape[1], P, Cc, 32, _lib.ptr(gamma_beta[0]),
                                                      _lib.ptr(gamma_beta[1]), 1e-6, _lib.ptr(sc), _lib.ptr(sh), _lib.stream()))
            return sc, sh
        work = torch.empty(B * ((P + 255) // 256) * 2 * Cc, device=dev, dtype=torch.float64)
        _lib.check(_lib.lib().ds_groupnorm_stats(_lib.ptr(x), B, P, Cc, 32, _lib.ptr(gamma_beta[0]),
                                                 _lib.ptr(gamma_beta[1]), 1e-6, _lib.ptr(work), _lib.ptr(sc),
                                                 _lib.ptr(sh), _lib.stream()))
        return sc, sh

    def _conv3(self, x, B, H, W, Cin, wb, gn=None, R=None, up=0):
        """3x3 conv, output H x W (input H/2 x W/2 if up == 1, 2H x 2W if up == 2).  gn = (scale, shift) ->
        GroupNorm + swish prologue."""
        w, b, w2, sc, wq = wb
        Cout = w.shape[0]
        out = torch.empty(B, H, W, Cout, device=x.device)
        if self.conv_precision == "f16x2" and H >= HALO_MIN_ROWS and up in (0, 1) and wq is not None:
            L = _lib.lib()
            part = torch.empty(B, L.ds_conv3x3_tiles(H, W), 2, Cout, device=x.device, dtype=torch.float64)
            _lib.check(L.ds_conv3x3_f16x2(_lib.ptr(x), _lib.ptr(wq), wq.numel(), sc, _lib.ptr(b), _lib.ptr(R), _lib.ptr(out),
                                          B, H, W, Cin, Cout, up, _lib.ptr(gn[0]) if gn is not None else None,
                                          _lib.ptr(gn[1]) if gn is not None else None, _lib.ptr(part), _lib.stream()))
            out._gn_part = part       # picked up by _gn() if a GroupNorm reads this tensor next
            return out
        kw = dict(bias=b, R=R, loader=_lib.LOAD_CONV2D,
                  pro=_lib.PRO_AFFINE_SWISH if gn is not None else _lib.PRO_NONE,
                  pro_scale=gn[0] if gn is not None else None, pro_shift=gn[1] if gn is not None else None,
                  Cin=Cin, H=H, Wd=W, up=up)
        if self.conv_precision == "f16x2" and Cout % 4 == 0:
            _lib.gemm(x, w2, out, B * H * W, Cout, 9 * Cin, split2=sc, conv_split=True, **kw)
        elif self.conv_precision in ("f16x2", "fp32"):
            _lib.gemm(x, w, out, B * H * W, Cout, 9 * Cin, **kw)
        else:
            raise ValueError("conv_precision must be 'f16x2' or 'fp32', got %r" % (self.conv_precision,))
        return out

    @staticmethod
    def _conv1(x, M, Cin, wb, R=None, gn=None, rows_per_sample=0):
        w, b = wb
        Cout = w.shape[0]
        out = torch.empty(M, Cout, device=x.device)
        _lib.gemm(x, w, out, M, Cout, Cin, bias=b, R=R,
                  pro=_lib.PRO_AFFINE if gn is not None else _lib.PRO_NONE,
                  pro_scale=gn[0] if gn is not None else None, pro_shift=gn[1] if gn is not None else None,
                  Cin=Cin, rows_per_sample=rows_per_sample)
        return out

    def _res(self, x, B, H, W, r):
        P = H * W
        h = self._conv3(x, B, H, W, r["cin"], r["c1"], gn=self._gn(x, B, P, r["cin"], r["n1"]))
        short = x if "nin" not in r else self._conv1(x, B * P, r["cin"], r["nin"])
        return self._conv3(h, B, H, W, r["cout"], r["c2"], gn=self._gn(h, B, P, r["cout"], r["n2"]), R=short)

    def _attn(self, x, B, H, W, a):
        Cc, P = a["c"], H * W
        Pp = (P + 31) // 32 * 32  # key axis padded to the GEMM's K granularity
        dev = x.device
        gn = self._gn(x, B, P, Cc, a["n"])
        qk = self._conv1(x, B * P, Cc, a["qk"], gn=gn, rows_per_sample=P)          # [B*P][2C]
        vT = torch.zeros(B, Cc, Pp, device=dev)                                     # [B][C][Pp]
        _lib.gemm(x, a["v"][0], vT, B * P, Cc, Cc, bias=a["v"][1], ldc=Pp, store=_lib.STORE_BATCH_T,
                  pro=_lib.PRO_AFFINE, pro_scale=gn[0], pro_shift=gn[1], Cin=Cc, rows_per_sample=P)
        S = torch.empty(B, P, Pp, device=dev)
        _lib.gemm(qk, qk.data_ptr() + 4 * Cc, S, P, P, Cc, lda=2 * Cc, ldw=2 * Cc, ldc=Pp, groups=B,
                  a_gstride=P * 2 * Cc, w_gstride=P * 2 * Cc, c_gstride=P * Pp)
        _lib.check(_lib.lib().ds_softmax_rows(_lib.ptr(S), B * P, P, Pp, float(int(Cc) ** (-0.5)), _lib.stream()))
        O = torch.empty(B * P, Cc, device=dev)
        _lib.gemm(S, vT, O, P, Cc, Pp, lda=Pp, ldw=Pp, ldc=Cc, groups=B,
                  a_gstride=P * Pp, w_gstride=Cc * Pp, c_gstride=P * Cc)
        return self._conv1(O, B * P, Cc, a["proj"], R=x).view(B, H, W, Cc)

    @torch.no_grad()
    def _decode_cl(self, q, B, H, W):
        """q: channels-last quant [B, H, W, C] -> mel [B, 1, 16H, 16W]."""
        pk, d = self._packed(), self.decoder
        Cz = q.shape[-1]
        h = self._conv1(q, B * H * W, Cz, pk["pq"]).view(B, H, W, -1)
        h = self._conv3(h, B, H, W, h.shape[-1], pk["conv_in"])
        r1, a1, r2 = pk["mid"]
        h = self._res(h, B, H, W, r1)
        h = self._attn(h, B, H, W, a1)
        h = self._res(h, B, H, W, r2)
        for lvl in reversed(range(d.num_resolutions)):
            u = pk["up"][lvl]
            for i, r in enumerate(u["block"]):
                h = self._res(h, B, H, W, r)
                if u["attn"]:
                    h = self._attn(h, B, H, W, u["attn"][i])
            if lvl != 0:
                H, W = 2 * H, 2 * W
                h = self._conv3(h, B, H, W, h.shape[-1], u["upsample"], up=1)
        Cc = h.shape[-1]
        gn = self._gn(h, B, H * W, Cc, pk["norm_out"])
        taps = torch.empty(B * H * W, 16, device=h.device)
        _lib.gemm(h, pk["conv_out"][0], taps, B * H * W, 9, Cc, ldc=16, pro=_lib.PRO_AFFINE_SWISH,
                  pro_scale=gn[0], pro_shift=gn[1], Cin=Cc, rows_per_sample=H * W)
        out = torch.empty(B, 1, H, W, device=h.device)
        _lib.check(_lib.lib().ds_stencil9(_lib.ptr(taps), 16, pk["conv_out"][1], _lib.ptr(out), B, H, W, _lib.stream()))
        return out

    @torch.no_grad()
    def _encode_cl(self, x, B, H, W):
        """x: f32 [B, H, W], the one-channel mel -> latent [B, H/16, W/16, C]
        (Encoder.forward, diffusionmodules/model.py:467-500, + quant_conv)."""
        pk, e = self._packed_encoder(), self.encoder
        w_in, b_in = pk["conv_in"]
        h = torch.empty(B, H, W, w_in.shape[0], device=x.device)
        part = torch.empty(B, _lib.lib().ds_conv3x3_c1_chunks(H, W), 2, w_in.shape[0], device=x.device, dtype=torch.float64)
        _lib.check(_lib.lib().ds_conv3x3_c1(_lib.ptr(x), _lib.ptr(w_in), _lib.ptr(b_in), _lib.ptr(h), B, H, W, w_in.shape[0],
                                            _lib.ptr(part), _lib.stream()))
        h._gn_part = part         # picked up by _gn(): the first ResnetBlock's GroupNorm needs no statistics pass
        for lvl in range(e.num_resolutions):
            dn = pk["down"][lvl]
            for i, r in enumerate(dn["block"]):
                h = self._res(h, B, H, W, r)
                if dn["attn"]:
                    h = self._attn(h, B, H, W, dn["attn"][i])
            if dn["downsample"] is not None:
                H, W = H // 2, W // 2
                h = self._conv3(h, B, H, W, h.shape[-1], dn["downsample"], up=2)    # stride-2, pad right/bottom
        r1, a1, r2 = pk["mid"]
        h = self._res(h, B, H, W, r1)
        h = self._attn(h, B, H, W, a1)
        h = self._res(h, B, H, W, r2)
        Cc = h.shape[-1]
        h = self._conv3(h, B, H, W, Cc, pk["conv_out"], gn=self._gn(h, B, H * W, Cc, pk["norm_out"]))
        return self._conv1(h, B * H * W, h.shape[-1], pk["quant"]).view(B, H, W, -1)

    # ---- reference-facing API -----------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x):
        """x f32[B, 1, 80, 848] -> (quant [B, 256, 5, 53], emb_loss, (perplexity, min_encodings, indices [B*265, 1]))
        (spec_codec/vqgan.py:54-60).  H and W must be multiples of 16 (four stride-2 stages)."""
        B, Cin, H, W = x.shape
        assert Cin == 1 and H % 16 == 0 and W % 16 == 0
        hs = []
        for s in range(0, B, self.decode_chunk):
            xc = x[s:s + self.decode_chunk, 0].float().contiguous()                # [b, H, W]: the one channel
            hs.append(self._encode_cl(xc, xc.shape[0], H, W))
        h = (torch.cat(hs, 0) if len(hs) > 1 else hs[0]).permute(0, 3, 1, 2).contiguous()
        return self.quantize(h)

    @torch.no_grad()
    def encode_latent(self, x):
        """quant_conv(encoder(x)) -> [B, 256, 5, 53] (the pre-quantisation latent; tests / partial pipelines)"""
        B, _, H, W = x.shape
        return self._encode_cl(x[:, 0].float().contiguous(), B, H, W).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def decode(self, quant):
        """quant f32[B, 256, 5, 53] -> f32[B, 1, 80, 848] (spec_codec/vqgan.py:62-65)."""
        B, Cz, H, W = quant.shape
        outs = []
        for s in range(0, B, self.decode_chunk):
            q = quant[s:s + self.decode_chunk].permute(0, 2, 3, 1).contiguous().float()
            outs.append(self._decode_cl(q, q.shape[0], H, W))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    @torch.no_grad()
    def forward(self, input):
        """mel -> (reconstruction, codebook loss) (spec_codec/vqgan.py:72-75)"""
        quant, diff, _ = self.encode(input)
        return self.decode(quant), diff

    @staticmethod
    def get_input(batch, k):
        """batch[k] [B, H, W] or [B, H, W, C] -> f32[B, C, H, W] (spec_codec/vqgan.py:77-82)"""
        x = batch[k]
        if x.dim() == 3:
            x = x[..., None]
        return x.permute(0, 3, 1, 2).contiguous().float()

    @torch.no_grad()
    def decode_tokens(self, tokens, H=5, W=53):
        """tokens i64[B, H*W] in sequence (column-major) order -> mel; fuses DALLE.decode_to_img's
        permuter + codebook lookup (dalle_spec.py:80-91) with decode()."""
        E = self.quantize.embedding.weight
        outs = []
        for s in range(0, tokens.shape[0], self.decode_chunk):
            tk = tokens[s:s + self.decode_chunk].contiguous()
            b = tk.shape[0]
            q = torch.empty(b, H, W, E.shape[1], device=tk.device)
            _lib.check(_lib.lib().ds_codebook_gather(_lib.ptr(tk), _lib.ptr(E), _lib.ptr(q), b, H, W, E.shape[1],
                                                     E.shape[0], _lib.stream()))
            outs.append(self._decode_cl(q, b, H, W))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]
