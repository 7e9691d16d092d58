"""Drop-in for the MelGAN generator (Diffsound/vocoder/modules.py:Generator :88-130), HIP-backed.

Same constructor, `model.N.*` state-dict keys (weight_g / weight_v / bias, as left by
torch.nn.utils.weight_norm, :18-23) and forward signature: mel f32[B, 80, T] in [0,1] ->
waveform f32[B, 1, 256*T].  Weight norm is folded once at pack time (the reference recomputes it on
every call); activations are channels-last [B, T, C]; every layer runs on the fp16 matrix cores with
the fp32-class 3-pass split (`conv_precision = "f16x2"`, default) or as gather-GEMMs on the exact-fp32 MFMA ("fp32"):
  Conv1d k7 (first layer)                   -> gather-GEMM, conv1d loader (conv_f16x2.hip)
  ConvTranspose1d(k=2r, s=r)                -> r two-tap convs over one staged tile: halo-tiled kernel (r = 8, conv1d_f16x2.hip),
                                               single pass with the weights in registers (r = 2, melgan_fused.hip)
  ResnetBlock, 128 / 256 channels           -> halo-tiled dilated k3 conv (conv1d_f16x2.hip) + ONE GEMM over [LReLU(h) | x] for the
                                               1x1 conv and the 1x1 shortcut
  ResnetBlock, 32 / 64 channels             -> ONE kernel per block (melgan_fused.hip): x read once, y written once
  final Conv1d(32->1, k7) + tanh            -> one pass (melgan_fused.hip)
(`conv_precision = "fp32"`: the same layers as gather-GEMMs on the exact-fp32 MFMA, the strict mode)
The whole batch goes through at once (the reference vocodes sample by sample,
evaluation/generate_samples_batch.py:183-187).
"""
import numpy as np
import torch
from torch import nn

from .. import _lib


class _WN(nn.Module):
    """Parameter container with the names torch's weight_norm leaves behind."""

    def __init__(self, shape_v, n_bias):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(n_bias))
        self.weight_g = nn.Parameter(torch.ones(shape_v[0], *([1] * (len(shape_v) - 1))))
        self.weight_v = nn.Parameter(torch.randn(*shape_v) * 0.02)

    def folded(self):
        """w = g * v / ||v||, norm over all dims but 0 (dim 0 = Cin for ConvTranspose1d)."""
        v = self.weight_v.detach().float()
        n = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
        return self.weight_g.detach().float() * v / n


def WNConv1d(cin, cout, k):
    return _WN((cout, cin, k), cout)


def WNConvTranspose1d(cin, cout, k):
    return _WN((cin, cout, k), cout)


class ResnetBlock(nn.Module):
    def __init__(self, dim, dilation=1):
        super().__init__()
        self.dilation = dilation
        # indices 2 and 4 hold the convs (0: LeakyReLU, 1: ReflectionPad1d, 3: LeakyReLU), :75-81
        self.block = nn.Sequential(nn.Identity(), nn.Identity(), WNConv1d(dim, dim, 3), nn.Identity(),
                                   WNConv1d(dim, dim, 1))
        self.shortcut = WNConv1d(dim, dim, 1)


class Generator(nn.Module):
    def __init__(self, input_size, ngf, n_residual_layers):
        super().__init__()
        ratios = [8, 8, 2, 2]
        self.ratios = ratios
        self.hop_length = int(np.prod(ratios))
        self.input_size = input_size
        mult = int(2 ** len(ratios))
        model = [nn.Identity(), WNConv1d(input_size, mult * ngf, 7)]
        for r in ratios:
            assert r % 2 == 0
            model += [nn.Identity(), WNConvTranspose1d(mult * ngf, mult * ngf // 2, r * 2)]
            for j in range(n_residual_layers):
                model += [ResnetBlock(mult * ngf // 2, dilation=3 ** j)]
            mult //= 2
        model += [nn.Identity(), nn.Identity(), WNConv1d(ngf, 1, 7), nn.Identity()]
        self.model = nn.Sequential(*model)
        self.n_residual_layers = n_residual_layers
        # "f16x2" (default): every layer on its best kernel of the 3-pass fp16 split (module docstring); "fp32": the strict
        # mode, every layer as a gather-GEMM on the exact-fp32 MFMA (three launches per ResnetBlock, polyphase GEMMs for the
        # transposed convs) -- also what the f16x2 mode falls back to for a shape none of its kernels is built for
        self.conv_precision = "f16x2"
        self._pk = None
        self._register_load_state_dict_pre_hook(lambda *a, **k: setattr(self, "_pk", None))

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def _packed(self):
        if self._pk is not None:
            return self._pk
        c1 = lambda m: (m.folded().permute(0, 2, 1).reshape(m.weight_v.shape[0], -1).contiguous(),
                        m.bias.detach().float().contiguous())   # [Cout][k][Cin]
        layers = list(self.model)
        first = layers[1]
        w = first.folded()                                       # [512, 80, 7]
        cpad = (w.shape[1] + 31) // 32 * 32
        wp = torch.zeros(w.shape[0], 7, cpad, device=w.device)
        wp[:, :, :w.shape[1]] = w.permute(0, 2, 1)
        pk = {"first": (wp.reshape(w.shape[0], -1).contiguous(), first.bias.detach().float().contiguous()),
              "cpad": cpad, "stages": []}
        i = 2
        for r in self.ratios:
            ct = layers[i + 1]
            w = ct.folded()                                      # [Cin, Cout, 2r]
            cin, cout, k = w.shape
            # phase p uses taps j = p (on x[s0]) and j = p + r (on x[s0-1]):  [r][Cout][2][Cin]
            wph = w.permute(2, 1, 0).reshape(2, r, cout, cin).permute(1, 2, 0, 3).reshape(r, cout, 2 * cin)
            st = {"r": r, "cin": cin, "cout": cout, "ct": (wph.contiguous(), ct.bias.detach().float().contiguous()),
                  "res": []}
            i += 2
            for _ in range(self.n_residual_layers):
                rb = layers[i]
                st["res"].append({"dil": rb.dilation, "c3": c1(rb.block[2]), "c1": c1(rb.block[4]),
                                  "sc": c1(rb.shortcut)})
                i += 1
            pk["stages"].append(st)
        last = layers[i + 2]
        wl = last.folded()                                       # [1, 32, 7] -> [7 taps][32]
        pk["last"] = (wl[0].permute(1, 0).contiguous(), float(last.bias.item()))
        # fp16 split planes of every GEMM weight (W * 2^s as hi + lo, out_scale = 2^-s) for conv_precision = "f16x2"
        sp = lambda w: _lib.split_f16x2(w.reshape(-1, w.shape[-1]))
        pk["first_s"] = sp(pk["first"][0])
        for st in pk["stages"]:
            st["ct_s"] = sp(st["ct"][0])
            # the phases' planes fragment-packed for the halo-tiled kernel (ds_convt1d_f16x2): [r][Cout/128][Cin/32][2 taps][..]
            r, cin, cout = st["r"], st["cin"], st["cout"]
            st["ct_q"] = None
            if cout % 128 == 0 and cin % 32 == 0:
                pl = st["ct_s"][0].view(2, r, cout, 2 * cin)
                st["ct_q"] = torch.cat([_lib.pack_conv_weights(pl[:, g].contiguous(), cout, cin, 2) for g in range(r)])
            for rb in st["res"]:
                for k in ("c3", "c1", "sc"):
                    rb[k + "_s"] = sp(rb[k][0])
                # [W2 | Ws] (one power-of-two scale for both) and b2 + bs for the one-GEMM block tail
                rb["tail_s"] = sp(torch.cat((rb["c1"][0], rb["sc"][0]), dim=1).contiguous())
                c = rb["c3"][0].shape[0]
                rb["c3_q"] = _lib.pack_conv_weights(rb["c3_s"][0], c, c, 3) if c % 128 == 0 else None
                rb["tail_b"] = (rb["c1"][1] + rb["sc"][1]).contiguous()
        self._pk = pk
        return pk

    def _mm(self, A, w, ws, out, M, N, K, **kw):
        """One gather-GEMM of the stack in the selected arithmetic (w: fp32 weights, ws: their split planes + scale)."""
        if self.conv_precision == "f16x2" and N % 4 == 0:
            return _lib.gemm(A, ws[0], out, M, N, K, split2=ws[1], conv_split=True, **kw)
        if self.conv_precision not in ("f16x2", "fp32"):
            raise ValueError("conv_precision must be 'f16x2' or 'fp32', got %r" % (self.conv_precision,))
        return _lib.gemm(A, w, out, M, N, K, **kw)

    @torch.no_grad()
    def forward(self, x, scale=1.0, shift=0.0):
        """x f32[B, 80, T] -> f32[B, 1, 256 T].  (scale, shift) lets the caller fold the
        (mel + 1) / 2 of generate_samples_batch.py:182 into the layout change."""
        pk = self._packed()
        B, Cm, T = x.shape
        dev = x.device
        x = x.contiguous().float()
        cpad = pk["cpad"]
        h = torch.empty(B, T, cpad, device=dev)
        _lib.check(_lib.lib().ds_mel_to_cl(_lib.ptr(x), _lib.ptr(h), B, Cm, T, cpad, float(scale), float(shift),
                                           _lib.stream()))
        w, b = pk["first"]
        c = w.shape[0]
        y = torch.empty(B, T, c, device=dev)
        self._mm(h, w, pk["first_s"], y, B * T, c, 7 * cpad, bias=b, loader=_lib.LOAD_CONV1D, Cin=cpad, Wd=T, taps=7, dil=1)
        h = y
        for st in pk["stages"]:
            r, cin, cout = st["r"], st["cin"], st["cout"]
            w, b = st["ct"]
            y = torch.empty(B, T * r, cout, device=dev)
            if self.conv_precision == "f16x2" and st["ct_q"] is not None:
                _lib.check(_lib.lib().ds_convt1d_f16x2(_lib.ptr(h), _lib.ptr(st["ct_q"]), st["ct_q"].numel(), st["ct_s"][1], _lib.ptr(b),
                                                       _lib.ptr(y), B, T, cin, cout, r, r // 2 + r % 2, 1, _lib.stream()))
            elif r == 2 and self.conv_precision == "f16x2" and _lib.lib().ds_melgan_convt2_ok(cin, cout):
                _lib.check(_lib.lib().ds_melgan_convt2(_lib.ptr(h), _lib.ptr(st["ct_s"][0]), r * cout * 2 * cin, st["ct_s"][1], _lib.ptr(b),
                                                       _lib.ptr(y), B, T, cin, cout, _lib.stream()))
            else:
                self._mm(h, w, st["ct_s"], y, B * T, cout, 2 * cin, bias=b, ldc=cout, loader=_lib.LOAD_CONVT1D,
                         pro=_lib.PRO_LRELU, store=_lib.STORE_CONVT, groups=r, w_gstride=cout * 2 * cin, Cin=cin, Wd=T,
                         ct_r=r, ct_p=r // 2 + r % 2, ct_tin=T)
            h, T = y, T * r
            for rb in st["res"]:
                M = B * T
                sc = torch.empty(B, T, cout, device=dev)
                if self.conv_precision == "f16x2":
                    # the whole block behind one entry: ONE kernel where it is built (h1 = None), else the dilated k3 conv
                    # into h1, then [LReLU(h1) | h] x [W2 | Ws]^T
                    one_pass = _lib.lib().ds_melgan_resblock_fused_ok(T, cout, rb["dil"])
                    h1 = None if one_pass else torch.empty(B, T, cout, device=dev)
                    w3, s3 = rb["c3_s"]
                    w2, osc = rb["tail_s"]
                    if not one_pass and rb["c3_q"] is not None and rb["dil"] <= 27:
                        # 128 / 256 channels: the k3 conv on the halo-tiled kernel, then the one-GEMM tail
                        L = _lib.lib()
                        _lib.check(L.ds_conv1d_k3_f16x2(_lib.ptr(h), _lib.ptr(rb["c3_q"]), rb["c3_q"].numel(), s3, _lib.ptr(rb["c3"][1]),
                                                        _lib.ptr(h1), B, T, cout, cout, rb["dil"], 1, _lib.stream()))
                        _lib.check(L.ds_melgan_resblock_tail(_lib.ptr(h1), _lib.ptr(h), _lib.ptr(w2), cout * 2 * cout, osc,
                                                             _lib.ptr(rb["tail_b"]), _lib.ptr(sc), M, cout, _lib.stream()))
                        h = sc
                        continue
                    _lib.check(_lib.lib().ds_melgan_resblock(_lib.ptr(h), _lib.ptr(w3), cout * 3 * cout, s3, _lib.ptr(rb["c3"][1]),
                                                             _lib.ptr(w2), cout * 2 * cout, osc, _lib.ptr(rb["tail_b"]), _lib.ptr(h1),
                                                             _lib.ptr(sc), B, T, cout, rb["dil"], _lib.stream()))
                else:
                    h1 = torch.empty(B, T, cout, device=dev)
                    self._mm(h, rb["c3"][0], rb["c3_s"], h1, M, cout, 3 * cout, bias=rb["c3"][1], loader=_lib.LOAD_CONV1D,
                             pro=_lib.PRO_LRELU, Cin=cout, Wd=T, taps=3, dil=rb["dil"])
                    self._mm(h, rb["sc"][0], rb["sc_s"], sc, M, cout, cout, bias=rb["sc"][1])
                    self._mm(h1, rb["c1"][0], rb["c1_s"], sc, M, cout, cout, bias=rb["c1"][1], R=sc, pro=_lib.PRO_LRELU)
                h = sc
        wl, bl = pk["last"]
        out = torch.empty(B, 1, T, device=dev)
        if self.conv_precision == "f16x2" and wl.shape[1] == 32:
            _lib.check(_lib.lib().ds_melgan_final(_lib.ptr(h), _lib.ptr(wl), bl, _lib.ptr(out), B, T, 32, _lib.stream()))
            return out
        taps = torch.empty(B * T, 8, device=dev)
        _lib.gemm(h, wl, taps, B * T, 7, wl.shape[1], ldc=8, pro=_lib.PRO_LRELU)
        _lib.check(_lib.lib().ds_stencil7_tanh(_lib.ptr(taps), 8, bl, _lib.ptr(out), B, T, _lib.stream()))
        return out
