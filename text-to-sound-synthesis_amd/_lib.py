"""ctypes binding of libdiffsound_hip.so (include/diffsound_hip.h).

There is no CPU fallback: if the shared object is missing or a call fails, this raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIFFSOUND_LIB") or os.path.join(_HERE, "libdiffsound_hip.so")   # env: A/B of two builds

# enums (diffsound_hip.h)
LOAD_DENSE, LOAD_CONV2D, LOAD_CONV1D, LOAD_CONVT1D = 0, 1, 2, 3
PRO_NONE, PRO_AFFINE, PRO_AFFINE_SWISH, PRO_LRELU = 0, 1, 2, 3
ACT_NONE, ACT_GELU2, ACT_TANH = 0, 1, 2
STORE_ROW, STORE_BATCH_T, STORE_CONVT, STORE_ATTN = 0, 1, 2, 3
(LP_ADALN1, LP_W_QKV, LP_B_QKV, LP_W_PROJ1, LP_B_PROJ1, LP_ADALN2, LP_W_Q2, LP_B_Q2, LP_W_KV2, LP_B_KV2,
 LP_W_PROJ2, LP_B_PROJ2, LP_LN2_G, LP_LN2_B, LP_W_FC1, LP_B_FC1, LP_W_FC2, LP_B_FC2, LP_COUNT) = range(19)

_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("bias", _vp), ("R", _vp), ("C", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32),
                ("lda", _i32), ("ldw", _i32), ("ldc", _i32), ("ldr", _i32),
                ("groups", _i32),
                ("a_gstride", _i64), ("w_gstride", _i64), ("c_gstride", _i64),
                ("loader", _i32), ("pro", _i32), ("act", _i32), ("store", _i32),
                ("pro_scale", _vp), ("pro_shift", _vp),
                ("rows_per_sample", _i32), ("Cin", _i32), ("H", _i32), ("Wd", _i32), ("up", _i32),
                ("taps", _i32), ("dil", _i32), ("ct_r", _i32), ("ct_p", _i32), ("ct_tin", _i32),
                ("f16_round", _i32), ("w3_plane", _i64), ("out_scale", _f),
                ("a_split", _i32), ("c_split", _i32), ("a_plane", _i64), ("c_plane", _i64),
                ("attn_kv", _vp), ("attn_heads", _i32), ("attn_nkey", _i32), ("attn_qplane", _i64)]


class DenoiserDesc(C.Structure):
    _fields_ = [("n_layer", _i32), ("n_embd", _i32), ("n_head", _i32), ("seq_len", _i32),
                ("cond_len", _i32), ("cond_dim", _i32), ("n_codes", _i32), ("n_steps", _i32),
                ("mlp_mult", _i32),
                ("tok_emb", _vp), ("pos_emb", _vp), ("lnf_g", _vp), ("lnf_b", _vp),
                ("w_logits", _vp), ("b_logits", _vp), ("sched", _vp)]


_PROTOS = {
    "ds_version": (C.c_int, []),
    "ds_last_error_string": (C.c_char_p, []),
    "ds_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "ds_gemm_force_tile": (None, [C.c_int]),
    "ds_gemm_f16x2": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "ds_conv2d_f16x2": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "ds_gemm_f16x2_multi": (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.c_int, _vp]),
    "ds_gemm_f16x2_auto_tile": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ds_gemm_f16x2_force_tile": (None, [C.c_int]),
    "ds_gemm_f16x2_set_balance_slots": (None, [C.c_int]),
    "ds_gemm_f16x2_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ds_denoiser_set_split_weights": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_f), _vp, _f]),
    "ds_embed": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_adaln": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ds_layernorm": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ds_attention": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attention_ex": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, _vp]),
    "ds_attention_f16x2": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_adaln_split": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ds_layernorm_split": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ds_attention_f16x2_split": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attn_nkey": (C.c_int, [C.c_int]),
    "ds_attention_f16x2_ready": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attn_pack_kv": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_embed_f16": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_layernorm_f16": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ds_l2norm_rows_f16": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "ds_sample_tail": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_sample_tail_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f, C.c_int, _vp]),
    "ds_q_sample": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_philox_uniforms": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_sample_tail_rng": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _f, C.c_int, _vp]),
    "ds_q_sample_rng": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_vq_argmin": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_loss_tail": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_loss_tail_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _f,
                                   C.c_int, _vp]),
    "ds_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "ds_layernorm_bwd_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ds_layernorm_bwd_sums": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "ds_colsum": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _i64, _i64, C.c_int, _vp]),
    "ds_colsum_ws": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _i64, _i64, C.c_int, _vp, _i64, _vp]),
    "ds_gelu2": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "ds_softmax_bwd_rows": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attention_bwd": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                   _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attention_bwd_f16x2": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                   _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_attention_bwd_f16x2_mon": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                   _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _vp, _vp]),
    "ds_embed_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_adamw": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, C.c_int, _vp]),
    "ds_adamw_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp]),
    "ds_amax": (C.c_int, [_vp, _i64, _vp, _vp]),
    "ds_pack_operand": (C.c_int, [_vp, C.c_int, C.c_int, _i64, _f, C.c_int, _vp, _i64, _vp, _i64, _vp, _i64, C.c_int, C.c_int, C.c_int,
                                  _vp, _vp, _vp]),
    "ds_pack_operand_tile_rows": (C.c_int, [C.c_int, C.c_int]),
    "ds_adamw_multi": (C.c_int, [_vp, C.c_int, _vp, _f, _f, _f, _f, _vp]),
    "ds_ema_multi": (C.c_int, [_vp, C.c_int, _f, _f, _vp]),
    "ds_grad_norm_multi": (C.c_int, [_vp, C.c_int, _vp, _i64, _f, _vp, _vp, _vp]),
    "ds_denoiser_create": (C.c_int, [C.POINTER(DenoiserDesc), C.POINTER(_vp), C.POINTER(_vp)]),
    "ds_denoiser_destroy": (None, [_vp]),
    "ds_denoiser_set_row_padding": (C.c_int, [_vp, C.c_int]),
    "ds_denoiser_rows_per_sample": (C.c_int, [_vp, C.c_int]),
    "ds_denoiser_workspace_bytes": (_i64, [_vp, C.c_int]),
    "ds_denoiser_kv_bytes": (_i64, [_vp, C.c_int]),
    "ds_denoiser_cond_kv": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "ds_denoiser_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "ds_denoiser_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f, _vp, _vp, _vp]),
    "ds_denoiser_step_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f, C.c_int, _vp, _vp, _vp]),
    "ds_denoiser_step_rng": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int, _f, C.c_int,
                                       _vp, _vp, _vp]),
    "ds_denoiser_sample_rng": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int, _f,
                                         C.c_int, _vp, _vp]),
    "ds_profile_enable": (C.c_int, [C.c_int]),
    "ds_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ds_profile_collect_n": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64), C.c_int]),
    "ds_codebook_gather": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_groupnorm_stats": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "ds_groupnorm_finish": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _f, _vp, _vp, _vp]),
    "ds_conv3x3_f16x2": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _vp, _vp, _vp, _vp]),
    "ds_conv3x3_tiles": (C.c_int, [C.c_int, C.c_int]),
    "ds_conv1d_k3_f16x2": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_melgan_resblock_tail": (C.c_int, [_vp, _vp, _vp, _i64, _f, _vp, _vp, C.c_int, C.c_int, _vp]),
    "ds_melgan_resblock": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, _i64, _f, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_softmax_rows": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _f, _vp]),
    "ds_conv3x3_c1": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ds_conv3x3_c1_chunks": (C.c_int, [C.c_int, C.c_int]),
    "ds_rows_times_matrix": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_rows_outer": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_stencil9": (C.c_int, [_vp, C.c_int, _f, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_melgan_resblock_fused_ok": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ds_convt1d_f16x2": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_melgan_convt2": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_melgan_convt2_ok": (C.c_int, [C.c_int, C.c_int]),
    "ds_melgan_final": (C.c_int, [_vp, _vp, _f, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ds_stencil7_tanh": (C.c_int, [_vp, C.c_int, _f, _vp, C.c_int, C.c_int, _vp]),
    "ds_mel_to_cl": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _vp]),
}
EXPORTED = tuple(_PROTOS)

_lib = None


def lib():
    """The loaded shared object (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdiffsound_hip.so is missing (%s). Build it with `python __graft_entry__.py` -- "
                "there is no CPU or PyTorch fallback for the Diffsound HIP path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)   # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class DiffsoundHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise DiffsoundHipError("libdiffsound_hip: rc=%d: %s" % (rc, lib().ds_last_error_string().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses anything that is not fp32/int64/f64
    contiguous device memory -- the kernels have no host path."""
    if t is None:
        return None
    if isinstance(t, int):   # an already-computed device address (sub-matrix views)
        return t
    if not t.is_cuda:
        raise DiffsoundHipError("tensor is not on a GPU: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise DiffsoundHipError("tensor must be contiguous")
    return t.data_ptr()


def ptr_off(t, elems):
    """Device address `elems` elements into a contiguous tensor (a column range of a fused projection, addressed in place
    with the tensor's row stride)."""
    return ptr(t) + elems * t.element_size()


def stream():
    return torch.cuda.current_stream().cuda_stream


def gemm_multi(descs, cfg):
    """ds_gemm_f16x2_multi: the products described by `descs` (gemm(..., desc_only=True); packed operands, plain row store, equal
    `groups`) as ONE grid of tile configuration cfg (0: 128x128, 1: 128x64, 3: 96x128)."""
    arr = (GemmDesc * len(descs))(*descs)
    check(lib().ds_gemm_f16x2_multi(arr, len(descs), cfg, stream()))


def gemm(A, W, C_out, M, N, K, *, bias=None, R=None, lda=None, ldw=None, ldc=None, ldr=None,
         groups=1, a_gstride=0, w_gstride=0, c_gstride=0, loader=LOAD_DENSE, pro=PRO_NONE,
         act=ACT_NONE, store=STORE_ROW, pro_scale=None, pro_shift=None, rows_per_sample=0,
         Cin=0, H=0, Wd=0, up=0, taps=0, dil=1, ct_r=0, ct_p=0, ct_tin=0, f16_round=0, split2=None,
         a_plane=0, c_plane=0, attn=None, w_plane=None, conv_split=False, desc_only=False):
    """split2: out_scale from split_f16x2(); W is its [2][N][K] fp16 split and the f16x2 kernel is used;
    a_plane / c_plane > 0 (f16x2 only): A and W are given / C is written as packed split planes (pack_planes())
    that many halves apart."""
    d = GemmDesc()
    d.A, d.W, d.bias, d.R, d.C = ptr(A), ptr(W), ptr(bias), ptr(R), ptr(C_out)
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else K
    d.ldw = ldw if ldw is not None else K
    d.ldc = ldc if ldc is not None else N
    d.ldr = ldr if ldr is not None else d.ldc
    d.groups, d.a_gstride, d.w_gstride, d.c_gstride = groups, a_gstride, w_gstride, c_gstride
    d.loader, d.pro, d.act, d.store = loader, pro, act, store
    d.pro_scale, d.pro_shift = ptr(pro_scale), ptr(pro_shift)
    d.rows_per_sample, d.Cin, d.H, d.Wd, d.up = rows_per_sample, Cin, H, Wd, up
    d.taps, d.dil, d.ct_r, d.ct_p, d.ct_tin = taps, dil, ct_r, ct_p, ct_tin
    d.f16_round = f16_round
    if split2 is not None and conv_split:        # conv-family loaders on the fp16 matrix cores (conv_f16x2.hip)
        d.w3_plane = w_plane if w_plane is not None else max(1, groups) * N * d.ldw
        d.out_scale = split2
        check(lib().ds_conv2d_f16x2(C.byref(d), stream()))
    elif split2 is not None:
        d.w3_plane = N * d.ldw
        d.out_scale = split2
        d.a_split, d.a_plane = int(a_plane > 0), a_plane
        if a_plane > 0:
            d.w3_plane = (N + 15) // 16 * 16 * d.ldw          # packed W: rows padded to 16
        d.c_split, d.c_plane = int(c_plane > 0), c_plane
        if w_plane is not None:
            d.w3_plane = w_plane
        if attn is not None:        # (kv images tensor or None, heads, nkey, q plane stride): STORE_ATTN
            d.attn_kv, d.attn_heads, d.attn_nkey, d.attn_qplane = ptr(attn[0]), attn[1], attn[2], attn[3]
        if desc_only:               # for gemm_multi(): the descriptor of this f16x2 product, not launched
            return d
        check(lib().ds_gemm_f16x2(C.byref(d), stream()))
    else:
        check(lib().ds_gemm(C.byref(d), stream()))
    return C_out


def pack_conv_weights(planes, Cout, Cin, taps):
    """[2][Cout][taps * Cin] fp16 planes of a conv's weights (K ordered [tap][channel], split_f16x2) -> the fragment-packed
    layout ds_conv3x3_f16x2 (taps = 9) / ds_conv1d_k3_f16x2 (taps = 3) load straight into MFMA B operands:
    [Cout/128][Cin/32][taps][2 planes][4 blocks of 32 output channels = the wave][2 k-steps][64 lanes][8 halves], where lane
    (hh = lane >> 5, l = lane & 31) of fragment (wave, ks) holds W[n = 128 nt + 32 wave + l][tap][32 slab + 16 ks + 8 hh + 0..7]."""
    assert Cout % 128 == 0 and Cin % 32 == 0
    w = planes.view(torch.int16).view(2, Cout // 128, 2, 2, 32, taps, Cin // 32, 2, 2, 8)     # pl nt wn j l tap ns ks hh e
    return w.permute(1, 6, 5, 0, 2, 3, 7, 8, 4, 9).contiguous().view(-1)


def pack_conv3x3_weights(planes, Cout, Cin):
    return pack_conv_weights(planes, Cout, Cin, 9)


def pack_planes(x2):
    """[2][R][K] fp16 planes -> the packed layout of include/diffsound_hip.h (ds_gemm_desc.a_split):
    [2][ceil(R/16)][K/32][16][4][8], chunk c of row r at position c ^ ((r >> 2) & 3); rows zero-padded to 16."""
    two, R, K = x2.shape
    assert two == 2 and K % 32 == 0
    R16 = (R + 15) // 16 * 16
    x = torch.zeros(2, R16, K, dtype=x2.dtype, device=x2.device)
    x[:, :R] = x2
    x = x.view(2, R16 // 16, 16, K // 32, 4, 8).permute(0, 1, 3, 2, 4, 5)       # [2][rg][kt][16 rows][4 chunks][8]
    r = torch.arange(16, device=x2.device)
    src = torch.arange(4, device=x2.device)[None, :] ^ ((r[:, None] >> 2) & 3)    # position p holds chunk p ^ swz(r)
    idx = src[None, None, None, :, :, None].expand(2, R16 // 16, K // 32, 16, 4, 8)
    return torch.gather(x, 4, idx).contiguous()


def unpack_planes(xp, R, K):
    """inverse of pack_planes: packed planes -> [2][R][K]"""
    xp = xp.reshape(2, -1, K // 32, 16, 4, 8)
    r = torch.arange(16, device=xp.device)
    src = torch.arange(4, device=xp.device)[None, :] ^ ((r[:, None] >> 2) & 3)    # the swizzle is an involution
    idx = src[None, None, None, :, :, None].expand(xp.shape)
    x = torch.gather(xp, 4, idx).permute(0, 1, 3, 2, 4, 5).reshape(2, -1, K)
    return x[:, :R].contiguous()


def attn_images(k2, v2, nkey):
    """Host-side mirror of the attention-ready K / V^T images (include/diffsound_hip.h, csrc/common.h): k2, v2 =
    [2][B][heads][Lk][64] fp16 planes -> [B][heads][4][nkey*64] (K hi | K lo | V^T hi | V^T lo), zero past Lk."""
    _, B, H, Lk, _ = k2.shape
    dev = k2.device
    key = torch.arange(Lk, device=dev)[:, None]
    d = torch.arange(64, device=dev)[None, :]
    koff = key * 64 + ((((d >> 3) ^ ((key >> 1) & 7))) << 3) + (d & 7)
    voff = d * nkey + (((key >> 3) ^ ((d >> 2) & 3)) << 3) + (key & 7)
    img = torch.zeros(B, H, 4, nkey * 64, dtype=torch.float16, device=dev)
    for pl in range(2):
        img[:, :, pl, koff.reshape(-1)] = k2[pl].reshape(B, H, -1)
        img[:, :, 2 + pl, voff.reshape(-1)] = v2[pl].reshape(B, H, -1)
    return img


def split_f16x2(w, packed=False):
    """fp32 [N][K] -> (int16 view of [2][N][K] fp16 planes of W * 2^s, out_scale = 2^-s).  s puts max|W| * 2^s in
    [2^13, 2^14): both planes stay in fp16's normal range for every weight that matters, nothing overflows.
    packed=True returns the planes in the packed layout (pack_planes) for GEMMs whose A operand is packed too."""
    import math
    w = w.detach().float()
    mx = float(w.abs().max())
    s = 0 if mx == 0.0 else 13 - math.floor(math.log2(mx))
    ws = w * (2.0 ** s)
    p0 = ws.to(torch.float16)
    p1 = (ws - p0.float()).to(torch.float16)
    planes = torch.stack((p0, p1)).contiguous()
    if packed:
        planes = pack_planes(planes)
    return planes.view(torch.int16), 2.0 ** (-s)
