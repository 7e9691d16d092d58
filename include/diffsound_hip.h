/* C ABI of libdiffsound_hip.so -- the MI355X (gfx950) kernels of the Diffsound generation path.
 *
 * The reference (yangdongchao/Text-to-sound-Synthesis) has no FFI / plugin boundary: its hot path
 * is stock torch ops called from Python nn.Modules.  This header therefore
 * defines the boundary a maintainer would bind instead of those op chains; each entry cites the
 * reference lines it replaces (paths relative to Diffsound/).  INTEGRATION.md shows the ctypes
 * binding and the module-level drop-ins.
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 / int64 data unless stated otherwise;
 * callers own all memory (outputs and workspaces included); calls are asynchronous on `stream`
 * (a hipStream_t, passed as void*); the return value is 0 on success, -1 for a rejected argument,
 * -2 for a failed launch, and ds_last_error_string() describes the last failure of the calling
 * thread.  No function allocates, frees or synchronises.  Activations are channels-last.
 * Compute entries keep no state between calls (a ds_denoiser handle only holds the caller's weight pointers).  The
 * exceptions are the TEST / MEASUREMENT switches -- ds_gemm*_force_tile, ds_gemm_f16x2_set_balance_slots, ds_profile_* --
 * which are process-global and must not be flipped while another thread or stream of the process is launching.
 */
#ifndef DIFFSOUND_HIP_H
#define DIFFSOUND_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* ds_stream_t; /* hipStream_t */

int ds_version(void);
const char* ds_last_error_string(void);

/* ---- gather-GEMM (fp32 MFMA): C = store(act(A(m,k) * W[n][k] + bias) + R) ----------------------
 * replaces nn.Linear (sound_synthesis/modeling/transformers/transformer_utils.py:31-36,75-82,
 * 248-253,345-348), Conv2d 1x1/3x3 (specvqgan/modules/diffusionmodules/model.py:92-226,570-671) and
 * Conv1d / ConvTranspose1d (vocoder/modules.py:72-127). */
enum { DS_LOAD_DENSE = 0, DS_LOAD_CONV2D = 1, DS_LOAD_CONV1D = 2, DS_LOAD_CONVT1D = 3 };
enum { DS_PRO_NONE = 0, DS_PRO_AFFINE = 1, DS_PRO_AFFINE_SWISH = 2, DS_PRO_LRELU = 3 };
enum { DS_ACT_NONE = 0, DS_ACT_GELU2 = 1, DS_ACT_TANH = 2 };
enum { DS_STORE_ROW = 0, DS_STORE_BATCH_T = 1, DS_STORE_CONVT = 2, DS_STORE_ATTN = 3 };

typedef struct ds_gemm_desc {
    const float* A;        /* activation base */
    const float* W;        /* [groups][N][ldw], K contiguous */
    const float* bias;     /* [N] or NULL */
    const float* R;        /* residual [M][ldr] (row-major store only) or NULL; may alias C */
    float* C;
    int32_t M, N, K;       /* per group; K % 32 == 0 */
    int32_t lda, ldw, ldc, ldr;
    int32_t groups;        /* >= 1; A/W/C advance by the strides below per group */
    int64_t a_gstride, w_gstride, c_gstride;
    int32_t loader, pro, act, store;
    const float* pro_scale; /* [samples][Cin] for DS_PRO_AFFINE* (GroupNorm folded to a*s + o) */
    const float* pro_shift;
    int32_t rows_per_sample; /* dense prologue / DS_STORE_BATCH_T: rows of one sample; ds_gemm_f16x2 with packed operands: lets
                                the dispatcher take the per-sample programs (see ds_gemm_f16x2_force_tile), 0 = never */
    int32_t Cin;             /* channels per tap (conv loaders; dense prologue: = K) */
    int32_t H, Wd;           /* conv2d: output H, W; conv1d / convT1d: Wd = output length / phase rows */
    int32_t up;              /* conv2d: 1 = source is (H/2, W/2), nearest-upsampled on the fly; 2 = stride-2 conv over a
                                (2H, 2W) source zero-padded right/bottom only (Downsample, model.py:60-77) */
    int32_t taps, dil;       /* conv1d (reflect padding) */
    int32_t ct_r, ct_p, ct_tin; /* convT1d polyphase: stride, padding, input length; groups = r */
    int32_t f16_round;       /* 1: round outputs (and GELU2 intermediates) to the fp16 grid (CLIP text tower) */
    int64_t w3_plane;        /* split kernels only: W holds 3 bf16 / 2 fp16 planes [N][ldw], w3_plane elements apart */
    float out_scale;         /* ds_gemm_f16x2 only: 2^-s undoing the exact power-of-two pre-scale of W */
    /* ds_gemm_f16x2 only -- "packed split planes" of X[R][K] (K % 32 == 0): two fp16 planes (hi, lo), each
       [ceil(R/16)][K/32][16 rows][4 chunks][8 halves] with 16-byte chunk c of row r stored at chunk c ^ ((r>>2)&3);
       element (r,k) at ((r/16)*(K/32) + k/32)*512 + (r%16)*32 + (((k/8)%4) ^ ((r/4)%4))*8 + k%8.
       a_split: A AND W are given in that layout (lda = ldw = K; planes a_plane / w3_plane halves apart), staged by
       LDS-DMA.  c_split: C is written in that layout with row length ldc (planes c_plane halves apart). */
    int32_t a_split, c_split;
    int64_t a_plane, c_plane;
    /* ds_gemm_f16x2, store = DS_STORE_ATTN (packed operands only): column n = which*heads*64 + head*64 + d (which 0 Q,
       1 K, 2 V; N = 1..3 times heads*64), row = sample*rows_per_sample + position (rows_per_sample >= 128).
       C = the Q planes (attn_qplane halves apart), attn_kv = the K / V^T images with attn_nkey key slots
       (see ds_attention_f16x2_ready). */
    void* attn_kv;
    int32_t attn_heads, attn_nkey;
    int64_t attn_qplane;
} ds_gemm_desc;

int ds_gemm(const ds_gemm_desc* d, ds_stream_t stream);
void ds_gemm_force_tile(int cfg); /* test hook: 0..2 pins the block tile, -1 = auto */
/* The same dense contraction (DS_LOAD_DENSE, DS_PRO_NONE) on the fp16 matrix cores with fp32-class accuracy:
 * W*2^s and A are split into two fp16 planes each (22 significant bits), three fp16 MFMA passes
 * a0b0 + a0b1 + a1b0 per k-step, epilogue multiplies by out_scale = 2^-s.  |A| must stay below 65504
 * (gemm_f16x2.hip).  groups > 1 (plain row store, no bias / residual): group g computes
 * A + g a_gstride (floats; HALVES for packed operands) times W + g w_gstride (halves, both planes) into C + g c_gstride
 * -- with a_gstride = w_gstride = K (row-major operands) resp. 16 K (packed operands: K / 32 k-tiles of 512 halves, and
 * lda = ldw = the FULL contraction length the planes were packed with) this is a split-K launch whose partial results the
 * caller sums (ds_colsum). */
int ds_gemm_f16x2(const ds_gemm_desc* d, ds_stream_t stream);
/* n <= 4 INDEPENDENT packed-operand products (a_split = 1, row store, no bias / residual / activation, the same `groups` each)
 * as ONE grid of tile configuration cfg (0: 128x128, 1: 128x64, 3: 96x128): the weight gradients dW = dY^T X of several
 * nn.Linear layers of a block (transformer_utils.py:31-36,75-82,248-253 under loss.backward()), each sized to one workgroup per
 * CU, side by side instead of back to back.  Every product gets the bits ds_gemm_f16x2 gives it with that tile forced. */
int ds_gemm_f16x2_multi(const ds_gemm_desc* descs, int n, int cfg, ds_stream_t stream);
/* the conv-family loaders in the same fp32-class 3-pass formulation: A fp32 (split while it is staged), W = the two fp16
   planes [groups][N][ldw] of W * 2^s from split_f16x2 (w3_plane halves apart, groups w_gstride apart), out_scale = 2^-s.
   d->loader: DS_LOAD_CONV2D (3x3 conv over a channels-last image: Cin, H, Wd, up; prologue none or GroupNorm affine +
   swish), DS_LOAD_CONV1D (taps, dil, reflect padding; prologue none / LeakyReLU), DS_LOAD_CONVT1D (groups = ct_r
   phases, DS_STORE_CONVT; vocoder/modules.py:104-111), DS_LOAD_DENSE (1x1 convs).  Bias, residual, row store; Cin % 32
   == 0, N % 4 == 0.  (A descriptor with loader 0 and the conv2d geometry H > 0, K == 9 Cin is taken as DS_LOAD_CONV2D.) */
int ds_conv2d_f16x2(const ds_gemm_desc* d, ds_stream_t stream);
/* cfg: -1 automatic (default): the per-sample ping-pong program (gemm_f16x2_ps.hip: one 288 x 256 tile of an 8-wave
   workgroup per (sample, 256 columns)) when the problem has packed operands, rows_per_sample in (240, 273] with
   M % rows_per_sample == 0, N % 256 == 0, K % 64 == 0, a row / packed / attention store and a grid that fills whole
   rounds of the 256 CUs (the denoiser's GEMMs at batch 64) -- or, for a row-major store, its LEADING samples that do, the
   rest following as a second launch (20 samples x 4096 columns: 16 + 4); else 128x128 (balanced launch for packed
   operands) / 128x64 / 64x64 / 96x128 (packed operands only: four waves side by side) tiles of a 4-wave workgroup by grid
   size (the rule for packed operands: gemm_f16x2.hip).  0 / 1 / 2 / 3 pin those four; 9 pins the per-sample program, 10 its
   half-tile form, for every problem they can compute, whatever the grid (tests).  Every cfg produces the same bits.
   PROCESS-GLOBAL test / measurement switch, like every *_force_tile, *_set_balance_slots and ds_profile_* entry:
   not for use while another thread or stream of the process is launching GEMMs. */
void ds_gemm_f16x2_force_tile(int cfg);
/* the tile configuration (0 .. 3 as above) an M x N packed-operand product in `groups` K-ranges takes when nothing is forced
   (pure arithmetic; the training step groups its weight gradients by it for ds_gemm_f16x2_multi) */
int ds_gemm_f16x2_auto_tile(int M, int N, int groups);
/* packed-operand launches that pick the 128x128 tile are balanced: 128x128 tiles over the leading rows that fill
   whole rounds of `slots` resident workgroups (default 512 = 256 CUs x 2), 64x64 tiles over the rest.  Test hook
   (process-global). */
void ds_gemm_f16x2_set_balance_slots(int slots);
/* The row partition a packed-operand launch of cfg 0 uses for an M x N product with the given store mode: rows
   [0, m_off) -> nbig main tiles, rows [m_off, M) -> nsmall tail tiles (0: one program over all rows).
   Pure arithmetic, no device work: the CPU test-suite checks the partition with it. */
int ds_gemm_f16x2_plan(int cfg, int M, int N, int store, int* m_off, int* nbig, int* nsmall);

/* ---- row kernels of the denoiser -------------------------------------------------------------- */
/* DalleMaskImageEmbedding.forward, sound_synthesis/modeling/embeddings/dalle_mask_image_embedding.py:36-58
 * out[m] = emb[tokens[m]] + pos[m % L]   (pos = height_emb[p // W] + width_emb[p % W], precombined) */
int ds_embed(const int64_t* tokens, const float* emb, const float* pos, float* out, int M, int L, int D,
             ds_stream_t stream);
/* AdaLayerNorm.forward, transformer_utils.py:145-149, with Linear(SiLU(Emb[t])) tabulated: table [T][2D] */
int ds_adaln(const float* x, float* y, int M, int L, int D, const float* table, const int64_t* t,
             ds_stream_t stream);
/* nn.LayerNorm(D) eps 1e-5 with affine, transformer_utils.py:205,346 */
int ds_layernorm(const float* x, float* y, int M, int D, const float* gamma, const float* beta,
                 ds_stream_t stream);
/* FullAttention / CrossAttention core, transformer_utils.py:43-58, 91-109 (head dim 64, Lk <= 288) */
int ds_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                 int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream);

/* causal = 1: key j visible to query i iff j <= i; f16_round = 1: scores, probabilities and outputs are
 * rounded to the fp16 grid (CLIP's nn.MultiheadAttention in fp16, clip/model.py:166-186) */
int ds_attention_ex(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                    int B, int heads, int Lq, int Lk, float scale, int causal, int f16_round, ds_stream_t stream);
/* the same attention (no mask, head dim 64) on the fp16 matrix cores with the f16x2 split: fp32-class results,
 * ~5x fewer MFMA cycles (attention_f16x2.hip); used by the denoiser in f16x2 mode */
int ds_attention_f16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                       int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream);
/* producers that write packed split planes for ds_gemm_f16x2 (a_split): yh / oh = 2 planes of ceil16(rows) * (D or
   ldo) halves, layout as described at ds_gemm_desc.a_split */
int ds_adaln_split(const float* x, void* yh, int M, int L, int D, const float* table, const int64_t* t,
                   ds_stream_t stream);
int ds_layernorm_split(const float* x, void* yh, int M, int D, const float* gamma, const float* beta,
                       ds_stream_t stream);
int ds_attention_f16x2_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, void* oh,
                             int ldo, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream);
/* Attention on "attention-ready" operands, the denoiser's default path: Q as two fp16 planes [B][heads][Lq][64]
 * (q_plane halves apart), K and V^T as per-(sample, head) LDS images  K hi | K lo | V^T hi | V^T lo  of nkey*64 halves
 * each (nkey = ds_attn_nkey(Lk); K[key][64] with 16-byte chunk c at c ^ ((key>>1)&7), V^T[d][nkey] with the 16-byte
 * chunk of keys 8c..8c+7 at c ^ ((d>>2)&3): csrc/common.h ds_attn_k_off / ds_attn_vt_off; rows of keys >= Lk zero).
 * Producers: ds_gemm_f16x2 with store = DS_STORE_ATTN (Q | K | V columns of the fused QKV projection, or Q alone),
 * ds_attn_pack_kv (fp32 K | V rows, e.g. the caption K/V).  Output: packed split planes as ds_attention_f16x2_split. */
int ds_attn_nkey(int Lk);
int ds_attention_f16x2_ready(const void* qh, long long q_plane, const void* kv_img, void* oh, int ldo, int B, int heads,
                             int Lq, int Lk, float scale, ds_stream_t stream);
int ds_attn_pack_kv(const float* kv, int ld, int v_col, void* img, int B, int heads, int Lk, ds_stream_t stream);
/* fp16-semantics row kernels of the CLIP text tower (sound_synthesis/modeling/modules/clip/model.py:150-157,
 * 341-354; embeddings/clip_text_embedding.py:46-88): fp32 storage, outputs rounded to the fp16 grid */
int ds_embed_f16(const int64_t* tokens, const float* emb, const float* pos, float* out, int M, int L, int D,
                 ds_stream_t stream);
int ds_layernorm_f16(const float* x, float* y, int M, int D, const float* gamma, const float* beta,
                     ds_stream_t stream);
int ds_l2norm_rows_f16(const float* x, float* y, int M, int D, ds_stream_t stream);

/* ---- one reverse-diffusion step's per-column tail --------------------------------------------
 * predict_start (diffusion_transformer.py:285-289) + top-r truncation (models/dalle_spec.py:158-174)
 * + q_posterior (:293-339) + log_sample_categorical (:359-368).
 * logits [B*L][K]; xt [B][L]; t [B]; u [B][K+1][L] uniforms; sched [8][T+1];
 * dbg_* optional [B][K+1][L] dumps (NULL to skip); trunc_r < 0 disables truncation. */
int ds_sample_tail(const float* logits, const int64_t* xt, const int64_t* t, const float* u, const float* sched,
                   int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc, float* dbg_post, int B, int L,
                   int K, int T, int initial, float trunc_r, ds_stream_t stream);

/* + top-k truncation ('top{k}p', dalle_spec.py:147-157): trunc_k > 0 keeps the k largest log-probs of a column
 * (exclusive with trunc_r >= 0) */
int ds_sample_tail_ex(const float* logits, const int64_t* xt, const int64_t* t, const float* u, const float* sched,
                      int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc, float* dbg_post, int B, int L,
                      int K, int T, int initial, float trunc_r, int trunc_k, ds_stream_t stream);

/* q_sample (diffusion_transformer.py:370-377): x_t ~ q(x_t | x_0) on token ids, u [B][K+1][L] uniforms; used by
 * sample()'s filter_ratio > 0 branch (:643-651) */
int ds_q_sample(const int64_t* x0, const int64_t* t, const float* u, const float* sched, int64_t* out_tokens, int B,
                int L, int K, int T, ds_stream_t stream);

/* ---- the same samplers with the noise drawn INSIDE the kernel -----------------------------------------------------
 * The reference draws torch.rand_like(logits) per step (log_sample_categorical, diffusion_transformer.py:359-368), so
 * what a caption draws depends on the batch it is in.  SURVEY.md section 8(e) asks for per-sample noise keyed by the
 * global caption index so that a shard reproduces the single-GPU draw (pattern: Codebook/evaluation/
 * generate_samples_caps.py:153,306 -- rank-sharded sampler, nothing collective in the loop).  These entries draw from a
 * counter-based Philox4x32-10 stream: the uniform of class c = 64 j + lane (the [MASK] class K is j = K / 64, lane 0)
 * at grid position pos of caption gids[b] in sampler call `call` is word (j & 3) of
 *     Philox(counter = (64 (j >> 2) + lane, pos | rng_stream << 16, call, gids[b]), key = (seed lo, seed hi))
 * mapped to [0, 1) as (word >> 8) * 2^-24; rng_stream = 0 for reverse steps, 1 for q_sample.  gids [B] (device, each
 * < 2^32), L < 65536.  ds_philox_uniforms writes that stream out as u [B][K+1][L] (tests: *_rng(...) == the u-path fed
 * with it).  Host mirror (numpy): text_to_sound_synthesis_amd.shard.caption_uniforms. */
int ds_philox_uniforms(const int64_t* gids, unsigned long long seed, int call, int rng_stream, float* u, int B, int L,
                       int K, ds_stream_t stream);
int ds_sample_tail_rng(const float* logits, const int64_t* xt, const int64_t* t, const int64_t* gids,
                       unsigned long long seed, int call, const float* sched, int64_t* out_tokens, int B, int L, int K,
                       int T, int initial, float trunc_r, int trunc_k, ds_stream_t stream);
int ds_q_sample_rng(const int64_t* x0, const int64_t* t, const int64_t* gids, unsigned long long seed, int call,
                    const float* sched, int64_t* out_tokens, int B, int L, int K, int T, ds_stream_t stream);

/* forward terms of the training loss (DiffusionTransformer._train_loss, diffusion_transformer.py:408-476), one value
 * per grid position [B][L]: kl = KL(true posterior || model posterior) (:439-440), nll = the t == 0 decoder term
 * (:446), kl_aux = KL(x_0 || p(x_0|x_t)) over the K classes (:462); logits [B*L][K] of the network at (x_t, t),
 * dbg_model_log_prob optional [B][K+1][L].  Forward value only; its gradient is ds_loss_tail_bwd below. */
int ds_loss_tail(const float* logits, const int64_t* x0, const int64_t* xt, const int64_t* t, const float* sched,
                 float* kl, float* nll, float* kl_aux, float* dbg_model_log_prob, int B, int L, int K, int T,
                 ds_stream_t stream);

/* d(sum over samples of vb_loss) / d logits for the loss above (pt [B] = the sampling probabilities of t; mask
 * weights for masked / unmasked x_t positions; auxiliary loss weight and its adaptive flag): dlogits [B*L][K].  The
 * first kernel of the training step's backward (modeling/train.py composes the rest from the entries below, the GEMMs
 * and ds_attention_bwd). */
int ds_loss_tail_bwd(const float* logits, const int64_t* x0, const int64_t* xt, const int64_t* t, const float* pt,
                     const float* sched, float* dlogits, int B, int L, int K, int T, float mask_weight_masked,
                     float mask_weight_other, float aux_weight, int adaptive, ds_stream_t stream);

/* ---- row / elementwise kernels of the training step (csrc/train.hip; scope row 8f-3, building blocks) --------
 * AdaLN (mode 0: table [T][2D], t [M/L]) / LayerNorm (mode 1: gamma) backward: dx, and dyxn = dy * xn for the scale
 * gradient (NULL to skip); D = 1024 */
int ds_layernorm_bwd(const float* x, const float* dy, float* dx, float* dyxn, int M, int L, int D, int mode,
                     const float* table, const int64_t* t, const float* gamma, ds_stream_t stream);
/* The backward with the scale / shift gradient sums folded in: part[G * chunks][2][D] receives, per chunk of <= 16 rows of a
 * group (G = M / L samples for mode 0, one group for mode 1; chunks = ds_layernorm_bwd_chunks(M, L, mode)), the column sums of
 * dy * xn and of dy; ds_colsum over a group's chunks (R = chunks, C = 2 D) gives [d scale | d shift].  No [M][D] dyxn matrix. */
int ds_layernorm_bwd_chunks(int M, int L, int mode);
int ds_layernorm_bwd_sums(const float* x, const float* dy, float* dx, float* part, int M, int L, int D, int mode,
                          const float* table, const int64_t* t, const float* gamma, int accumulate, ds_stream_t stream);
/* out[g][c] (+)= sum over R rows of x[(g*gstride) + r*ld + c]: bias / scale / per-sample AdaLN gradients */
int ds_colsum(const float* x, float* out, int G, int R, int C, long long ld, long long gstride, int accumulate,
              ds_stream_t stream);
/* the same sums for tall inputs (R in the thousands): rows are summed in up to 64 chunks by separate workgroups into
 * `work` (>= G * 64 * C floats is always enough), then the chunks are added in a fixed order -- deterministic */
int ds_colsum_ws(const float* x, float* out, int G, int R, int C, long long ld, long long gstride, int accumulate,
                 float* work, long long work_floats, ds_stream_t stream);
/* GELU2: dy == NULL -> out = x * sigmoid(1.702 x); else out = dy * d/dx of that */
int ds_gelu2(const float* x, const float* dy, float* out, long long n, ds_stream_t stream);
/* in place dS = scale * P * (dP - rowsum(dP * P)) over the first n columns of each row (attention backward) */
int ds_softmax_bwd_rows(const float* P, float* dP, int rows, int n, int ld, float scale, ds_stream_t stream);
/* Backward of ds_attention (FullAttention / CrossAttention cores, transformer_utils.py:43-58, :91-109) by tile-wise
 * recomputation -- the probabilities are never stored: given Q, K, V, the forward's O and dO it writes dQ, dK, dV.
 * Same addressing as ds_attention (rows b*L + pos, head h at columns h*64.., any row strides: column ranges of fused
 * projections work in place); exact-fp32 MFMA; Lq, Lk <= 288.  stats: 2 * B * heads * ceil32(Lq) floats of workspace
 * (per query: log-sum-exp of its scaled scores and delta = dO . O). */
int ds_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                     const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                     float* stats, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream);
/* The same backward with every tile product on the fp16 matrix cores (operands split into fp16 hi + lo on the fly, three
 * v_mfma_f32_32x32x16_f16 passes, fp32 accumulate: fp32-class, 5.3x less matrix-pipe time than the exact-fp32 MFMA); |dO|
 * must stay below 65504 (the training step's loss scale puts it at <= 2^13).  The "f16x2" training backend's choice. */
int ds_attention_bwd_f16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                           const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                           float* stats, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream);
/* ... with the training step's saturation monitor folded in (round 6): *amax -- a device float the caller zeroes when it
 * reads it; the kernels only ever raise it -- takes max |dO * do_scale|, the operand that carries the step's loss scale into
 * these kernels' fp16 splits (replaces the separate ds_amax over dO of rounds 3-5).  do_scale: a positive power of two, this
 * call's own scale on top of the loss scale: dO * do_scale is what is split, the dQ / dK / dV stores take it out again (exact).  dS = scale P (dP - delta), which exists only
 * in registers and has no fixed relation to dO (it can exceed it by |V| x 8, or sit 2^-20 under it where the probabilities
 * are near-uniform), is normalised per wave by an exact power of two before ITS split in all three entry points of the f16x2
 * form -- it can neither saturate nor fall into fp16's subnormal range.  Reference: the same loss.backward() lines as above. */
int ds_attention_bwd_f16x2_mon(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                               const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                               float* stats, int B, int heads, int Lq, int Lk, float scale, float do_scale, float* amax,
                               ds_stream_t stream);
/* part[ks][g][b][d] = sum over k in [256 ks, 256 ks + 256) of x[g][b][k] * W[g][k][d]:  B <= 32 rows times G ROW-MAJOR matrices
 * [K][D] (K % 256 == 0), read once from where they lie -- the AdaLN backward's (d modulation) x linear.weight for all modules
 * (what autograd computes for AdaLayerNorm.linear's input, transformer_utils.py:145-147).  The caller adds the K / 256 partial
 * results in a fixed order (ds_colsum over part as [K / 256][G * B * D]). */
int ds_rows_times_matrix(const float* x, const float* W, float* part, int G, int B, int K, int D, ds_stream_t stream);
/* out[g][n][d] = sum over b < B of a[g][b][n] * s[g][b][d]  (a [G][B][N], s [G][B][D], out [G][N][D]; B <= 32, D % 4 == 0):
 * the AdaLN backward's d linear.weight = (d modulation)^T silu(emb(t_b)) for all modules in one output-bound pass (what
 * autograd computes for AdaLayerNorm.linear.weight, transformer_utils.py:145-147). */
int ds_rows_outer(const float* a, const float* s, float* out, int G, int B, int N, int D, ds_stream_t stream);
/* d emb[tokens[m]] += dx[m] (atomic) */
int ds_embed_bwd(const float* dx, const int64_t* tokens, float* demb, int M, int D, int rows, ds_stream_t stream);
/* fused AdamW update (torch.optim.AdamW semantics), step >= 1 */
int ds_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, ds_stream_t stream);
/* The same update with the iteration's scalars in device memory, hyper = { lr, 1 - beta1^step, sqrt(1 - beta2^step),
 * grad_scale } (g is multiplied by grad_scale first: the clip coefficient / the inverse loss scale): a captured
 * hipGraph replays fixed kernel arguments, so what changes per iteration is read through this pointer. */
int ds_adamw_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2,
                 float eps, float weight_decay, ds_stream_t stream);
/* *out = max(*out, max_i |x[i]|)  (the caller zeroes *out; calibration of the training step's loss scale) */
int ds_amax(const float* x, long long n, float* out, ds_stream_t stream);
/* ONE pass over fp32 X[rows][ld_src] (cols valid, cols % 32 == 0) that writes everything the three GEMMs of an nn.Linear
 * in the training step (y = x W^T, dX = dY W, dW = dY^T X: engine/solver_spec.py:308-331 over
 * diffusion_transformer.py:408-476) need of it -- any subset of:
 *   dst_row      packed split planes (hi | lo, plane_row halves apart) of scale * X as a [rows][cols] GEMM operand;
 *   dst_t        packed split planes (plane_t apart) of scale * X^T as a [cols][rows_pad] operand: the contraction index
 *                (X's rows) zero-padded to rows_pad (% 32 == 0; split-K launches need a multiple of 32 * groups); with
 *                t_cols > 0 the destination is [cols][t_cols] and this matrix fills its k-range [t_col0, t_col0 + rows_pad)
 *                (t_col0, t_cols % 32 == 0; 0, 0 = the matrix is the whole destination) -- row ranges of the ROW form are
 *                reached by offsetting dst_row by whole 16-row groups: the parts of a fused projection weight need no
 *                concatenated copy;
 *   colsum_part  [ds_pack_operand_tile_rows(rows, rows_pad)][cols] per-64-row column sums of X (after the prologue, BEFORE
 *                `scale`: a gradient's scale is its site's own power of two, which the bias gradient does not carry), the
 *                first ceil(rows / 64) rows of which ds_colsum adds up in a fixed order (bias gradients; no atomics);
 *   amax         *amax = max(*amax, max |scale * X|) (atomicMax on the bit pattern; the caller zeroes it);
 * with an elementwise prologue: DS_PACK_GELU2 (X := gelu2(X): GELU2 of transformer_utils.py:111-115 between the MLP's
 * linears) or DS_PACK_GELU2_BWD (X := X * gelu2'(aux), aux fp32 [rows][ld_aux]: its backward).  scale: a power of two. */
enum { DS_PACK_PLAIN = 0, DS_PACK_GELU2 = 1, DS_PACK_GELU2_BWD = 2 };
int ds_pack_operand(const float* src, int rows, int cols, long long ld_src, float scale, int pro, const float* aux,
                    long long ld_aux, void* dst_row, long long plane_row, void* dst_t, long long plane_t, int rows_pad,
                    int t_col0, int t_cols, float* colsum_part, float* amax, ds_stream_t stream);
int ds_pack_operand_tile_rows(int rows, int rows_pad);
/* torch.optim.AdamW (configs/caps.yaml:111-115) on many parameter tensors at once: `tensors` = HOST array of n_tensors
 * records { p, g, m, v (device pointers), n (int64 element count) } = 5 x 8 bytes each; 64 tensors per launch, their
 * descriptors passed by value in the kernel arguments (graph-capturable, no table in device memory); hyper and the
 * arithmetic as for ds_adamw_dev. */
int ds_adamw_multi(const void* tensors, int n_tensors, const float* hyper, float beta1, float beta2, float eps,
                   float weight_decay, ds_stream_t stream);
/* engine/clip_grad_norm.py:8-29 -> torch.nn.utils.clip_grad_norm_: total = the L2 norm over ALL gradient tensors and coef =
 * min(1, max_norm / (total + 1e-6)) in device memory (ds_adamw_multi reads it as hyper[3]; nothing comes back to the host):
 * `tensors` = HOST array of n_tensors records { g (device pointer, fp32), n (int64 element count) }; part = workspace of at
 * least sum ceil(n_i / 4096) doubles; partial sums per 4096-element chunk, added in a fixed order in double. */
int ds_grad_norm_multi(const void* tensors, int n_tensors, double* part, long long part_len, float max_norm, float* total,
                       float* coef, ds_stream_t stream);
/* engine/ema.py:40-56 (EMA.update: ema = ema * decay + current * (1 - decay) for every entry of the state dict) as one pass:
 * `tensors` = HOST array of n_tensors records { ema, current (device pointers, fp32), n (int64 element count) } = 3 x 8 bytes
 * each; 96 tensors per launch, descriptors by value (graph-capturable).  The reference's expression term for term: two
 * products and a sum, each rounded to fp32; one_minus_decay = the host's (1 - decay), which is what the reference multiplies by. */
int ds_ema_multi(const void* tensors, int n_tensors, float decay, float one_minus_decay, ds_stream_t stream);

/* ---- the whole denoiser (Text2ImageTransformer.forward, transformer_utils.py:421-443) ---------- */
enum {  /* per-layer device pointers, layer-major: ptrs[layer * DS_LP_COUNT + slot] */
    DS_LP_ADALN1 = 0,  /* [T][2D]  ln1 table   */
    DS_LP_W_QKV,       /* [3D][D]  attn1 query|key|value */
    DS_LP_B_QKV,       /* [3D] */
    DS_LP_W_PROJ1, DS_LP_B_PROJ1,
    DS_LP_ADALN2,      /* [T][2D]  ln1_1 table */
    DS_LP_W_Q2, DS_LP_B_Q2,
    DS_LP_W_KV2,       /* [2D][Dc] attn2 key|value */
    DS_LP_B_KV2,
    DS_LP_W_PROJ2, DS_LP_B_PROJ2,
    DS_LP_LN2_G, DS_LP_LN2_B,
    DS_LP_W_FC1, DS_LP_B_FC1, DS_LP_W_FC2, DS_LP_B_FC2,
    DS_LP_COUNT
};
typedef struct ds_denoiser_desc {
    int32_t n_layer, n_embd, n_head, seq_len, cond_len, cond_dim, n_codes, n_steps, mlp_mult;
    const float* tok_emb;   /* [n_codes+1][D] */
    const float* pos_emb;   /* [L][D] */
    const float* lnf_g;     /* to_logits.0 */
    const float* lnf_b;
    const float* w_logits;  /* [n_codes][D] */
    const float* b_logits;
    const float* sched;     /* [8][n_steps+1] */
} ds_denoiser_desc;
typedef struct ds_denoiser ds_denoiser;

int ds_denoiser_create(const ds_denoiser_desc* desc, const void* const* layer_ptrs, ds_denoiser** out);
void ds_denoiser_destroy(ds_denoiser* h);
int64_t ds_denoiser_workspace_bytes(const ds_denoiser* h, int B);
int64_t ds_denoiser_kv_bytes(const ds_denoiser* h, int B);
/* Switch the denoiser's per-step GEMMs to the split kernel.  mode DS_SPLIT_F16X2 = f16x2, DS_SPLIT_NONE = back to fp32
 * MFMA (the strict mode).  split[layer*DS_LP_COUNT + slot] holds the packed plane-split of that slot's weight for the
 * DS_LP_W_* slots QKV, PROJ1, Q2, PROJ2, FC1, FC2 (other slots NULL), out_scales[same index] its 2^-s;
 * w_logits_split / logits_scale likewise for to_logits.1.  (Value 1 was the 6-pass bf16 split of rounds 1-4: removed.) */
enum { DS_SPLIT_NONE = 0, DS_SPLIT_F16X2 = 2 };
int ds_denoiser_set_split_weights(ds_denoiser* h, int mode, const void* const* split, const float* out_scales,
                                  const void* w_logits_split, float logits_scale);
/* Padded-row mode of ds_denoiser_step(_ex) (default on): in f16x2 mode, at batch sizes whose GEMMs run the per-sample
 * program (B = 64), every sample occupies 272 rows = 17 packed row groups of the step's activation matrices instead of
 * seq_len = 265 -- tiles start on a group boundary and their ninth block row is 16 rows on the 16x16x32 MFMA (-5 % MFMA
 * work).  The 7 extra rows per sample are zero embeddings that stay finite, are never attention keys, and are skipped by
 * the sampler.  Tokens agree with the unpadded step except on exact near-ties (rows 256..264 of a sample are summed in
 * another order: ~1e-7 relative).  0 switches it off (A/B, bit-for-bit comparisons across batch sizes). */
int ds_denoiser_set_row_padding(ds_denoiser* h, int on);
/* rows per sample of the activation matrices ds_denoiser_step(_ex) would use at batch B: seq_len, or 272 in padded-row mode */
int ds_denoiser_rows_per_sample(const ds_denoiser* h, int B);
/* cross-attention K/V depend only on the caption: computed once per batch (CrossAttention.key/value,
 * transformer_utils.py:96,98).  cond [B][Lc][Dc] -> kv [n_layer][B*Lc][2D] */
int ds_denoiser_cond_kv(const ds_denoiser* h, const float* cond, int B, float* kv, ds_stream_t stream);
/* logits_layout 0: [B*L][K]; 1: [B][K][L] (the reference's output layout) */
int ds_denoiser_forward(const ds_denoiser* h, const int64_t* tokens, const int64_t* t, const float* kv, int B,
                        void* workspace, float* logits, int logits_layout, ds_stream_t stream);
/* forward + ds_sample_tail: tokens_in -> tokens_out for timestep vector t */
int ds_denoiser_step(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t, const float* kv,
                     const float* u, int B, int initial, float trunc_r, void* workspace, int64_t* tokens_out,
                     ds_stream_t stream);

/* the same with the posterior's own timestep vector t_post (NULL = t; sample_fast's skip-step sampler,
 * diffusion_transformer.py:796-803) and top-k truncation */
int ds_denoiser_step_ex(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t, const int64_t* t_post,
                        const float* kv, const float* u, int B, int initial, float trunc_r, int trunc_k,
                        void* workspace, int64_t* tokens_out, ds_stream_t stream);

/* ds_denoiser_step_ex with the noise drawn in the sampler kernel (ds_sample_tail_rng): Philox call index `call` */
int ds_denoiser_step_rng(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t, const int64_t* t_post,
                         const float* kv, const int64_t* gids, unsigned long long seed, int call, int B, int initial,
                         float trunc_r, int trunc_k, void* workspace, int64_t* tokens_out, ds_stream_t stream);

/* A whole reverse chain (DiffusionTransformer.sample's loop, diffusion_transformer.py:639-641; sample_fast's, :790-804)
 * enqueued without returning to the host: n_calls steps, t_steps = DEVICE i64[n_calls][2][B] (per call the network's
 * timestep vector, then the posterior's), tokens [B][seq_len] in: start state / out: result, tokens_tmp same size,
 * call k draws Philox call index call0 + k, initial != 0: the start state is all-[MASK]. */
int ds_denoiser_sample_rng(const ds_denoiser* h, int64_t* tokens, int64_t* tokens_tmp, const int64_t* t_steps,
                           int n_calls, const float* kv, const int64_t* gids, unsigned long long seed, int call0, int B,
                           int initial, float trunc_r, int trunc_k, void* workspace, ds_stream_t stream);

/* per-launch HIP-event timing of the denoiser's GEMM launches (measurement only, bench.py) */
int ds_profile_enable(int on);
/* arrays of 4, indexed by GEMM program (0: 128x128, 1: 128x64, 2: 64x64 tiles; 3: the per-sample 288x256 ping-pong
   program of the f16x2 mode) */
int ds_profile_collect(double* ms, double* flops, int64_t* launches);
/* the same for the first n <= 5 programs: 4 = the per-sample program on HALF tiles (144 | 128 x 256: the grids full tiles
   cannot fill, e.g. batch 32); ds_profile_collect folds it into entry 3 */
int ds_profile_collect_n(double* ms, double* flops, int64_t* launches, int n);

/* ---- SpecVQGAN decoder / MelGAN helpers ------------------------------------------------------- */
/* ColumnMajor(reverse) + get_codebook_entry (permuter.py:31-55, quantize.py:88-103) -> [B][H][W][C] */
int ds_codebook_gather(const int64_t* tokens, const float* codebook, float* out, int B, int H, int W, int C,
                       int K, ds_stream_t stream);
/* VectorQuantizer.forward's nearest-code search (vqvae/quantize.py:46-53): z [M][C] encoder output rows,
 * ze [M][K] = z E^T (ds_gemm), ee [K] = row norms^2 of the codebook -> idx [M] (first minimum), dmin [M] optional */
int ds_vq_argmin(const float* z, const float* ze, const float* ee, int64_t* idx, float* dmin, int M, int C, int K,
                 ds_stream_t stream);
/* GroupNorm(groups, C, eps) statistics of x [B][P][C] folded into per-(b,c) scale/shift
 * (model.py:34-35); work >= B*ceil(P/256)*2*C doubles */
int ds_groupnorm_stats(const float* x, int B, int P, int C, int groups, const float* gamma, const float* beta,
                       float eps, double* work, float* scale, float* shift, ds_stream_t stream);
/* its second half alone: part = [B][nchunk][2][C] double partial sums (sum | sum of squares per channel over a chunk of
 * the P pixels) produced elsewhere -- ds_conv3x3_f16x2 writes them per output tile (nchunk = ds_conv3x3_tiles(H, W)) */
int ds_groupnorm_finish(const double* part, int B, int nchunk, int P, int C, int groups, const float* gamma,
                        const float* beta, float eps, float* scale, float* shift, ds_stream_t stream);
/* The decoder's hot 3x3 convolutions as ONE halo-tiled kernel (csrc/conv3x3_f16x2.hip): ResnetBlock conv1 / conv2 with
 * the preceding GroupNorm + swish as a prologue (specvqgan/modules/diffusionmodules/model.py:92-151) and the Upsample
 * conv (:37-52: nearest-2x, then conv).  x [B][H][W][Cin] channels-last fp32 ([B][H/2][W/2][Cin] when up = 1); w2 = the
 * two fp16 planes of W * 2^s (split_f16x2 of W[Cout][9 Cin], K ordered [tap][channel]) in the fragment-packed layout
 * [Cout/128][Cin/32][9][2 planes][2][2][2][64 lanes][8]: lane (hh = lane >> 5, l = lane & 31) of fragment (wn, j, ks) holds
 * W[128 nt + 64 wn + 32 j + l][tap][32 slab + 16 ks + 8 hh + 0..7] (_lib.pack_conv3x3_weights), w_halves = 2 Cout 9 Cin halves
 * in all; out_scale = 2^-s; y = conv(act(x)) + bias (+ residual) [B][H][W][Cout]; stride 1, zero padding 1 (in the activated domain, as
 * the reference pads after the activation).  pro_scale / pro_shift [B][Cin]: GroupNorm folded to a * s + o, followed by
 * swish; NULL = no prologue.  gn_part != NULL: also writes the GroupNorm partial sums of y, [B][ds_conv3x3_tiles(H, W)]
 * [2][Cout] doubles, for ds_groupnorm_finish.  Cin % 32 == 0, Cout % 128 == 0; 3-pass fp16 split, fp32 accumulate. */
int ds_conv3x3_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias,
                     const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int up,
                     const float* pro_scale, const float* pro_shift, double* gn_part, ds_stream_t stream);
int ds_conv3x3_tiles(int H, int W);   /* 4 x 32 pixel tiles per image */
/* Dilated k = 3 Conv1d with ReflectionPad1d(dil) (MelGAN ResnetBlock's first conv, vocoder/modules.py:75-78), halo-tiled like
 * ds_conv3x3_f16x2:  y[b][t][n] = bias[n] + 2^-s sum_{j<3} sum_c W[n][j][c] act(x[b][reflect(t + (j - 1) dil)][c]),  act =
 * LeakyReLU(0.2) if lrelu.  x [B][T][Cin], y [B][T][Cout] channels-last fp32; w2 = the fp16 planes of W * 2^s fragment-packed
 * (_lib.pack_conv_weights(.., taps = 3); w_halves = 2 * Cout * 3 * Cin); out_scale = 2^-s.  Cin % 32 == 0, Cout % 128 == 0,
 * 0 < dil <= 27. */
int ds_conv1d_k3_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias, float* y, int B,
                       int T, int Cin, int Cout, int dil, int lrelu, ds_stream_t stream);
/* MelGAN ResnetBlock tail (vocoder/modules.py:72-85) in one contraction over K = 2 C: y = W2 LReLU(h) + Ws x + bias, h = the
 * block's dilated k3 conv output [M][C], x = the block input [M][C] (channels-last rows), w = the two fp16 planes of
 * [W2 | Ws] * 2^s ([C][2 C] row-major, w_plane halves apart, split_f16x2), out_scale = 2^-s, bias = b2 + bs.  Replaces the
 * shortcut GEMM + the 1x1 GEMM with residual: the shortcut tensor never goes through HBM.  C % 32 == 0. */
int ds_melgan_resblock_tail(const float* h, const float* x, const void* w, long long w_plane, float out_scale,
                            const float* bias, float* y, int M, int C, ds_stream_t stream);
/* The whole ResnetBlock behind one entry (SURVEY.md section 8b's `ds_melgan_resblock`): y = shortcut(x) +
 * conv1x1(LReLU(conv_k3_dil(reflect_pad_dil(LReLU(x))))).  x, y [B][T][C] channels-last fp32; w3 = the fp16 planes of the
 * weight-norm-folded k3 weights * 2^s ([C][3 C], K ordered [tap][channel]; w3_plane halves apart; w3_scale = 2^-s), b3 [C];
 * wt / wt_plane / wt_scale / bt = ds_melgan_resblock_tail's operands.  C % 32 == 0, 0 < dil < T.
 *   h != NULL: two launches -- the dilated k3 conv into the scratch tensor h ([B][T][C], caller-owned), then
 *              ds_melgan_resblock_tail;
 *   h == NULL: ONE single-pass kernel (x read once, y written once; the intermediate stays in registers) -- built for the
 *              shapes ds_melgan_resblock_fused_ok() accepts (C = 32: T % 128 == 0, dil <= 16; C = 64: T % 256 == 0, dil <= 9), an
 *              error elsewhere. */
int ds_melgan_resblock(const float* x, const void* w3, long long w3_plane, float w3_scale, const float* b3, const void* wt,
                       long long wt_plane, float wt_scale, const float* bt, float* h, float* y, int B, int T, int C,
                       int dil, ds_stream_t stream);
int ds_melgan_resblock_fused_ok(int T, int C, int dil);   /* 1: ds_melgan_resblock(h = NULL) is available for this block */
/* ConvTranspose1d(k = 2 r, stride r, padding pad) in polyphase form on the halo-tiled kernel (MelGAN's stride-8 upsampling layers,
 * vocoder/modules.py:104-113):  y[b][(q + e_p) r + p - pad][n] = bias[n] + 2^-s sum_c (W[p][n][0][c] act(x[b][q + e_p][c]) +
 * W[p][n][1][c] act(x[b][q + e_p - 1][c])),  e_p = p < pad, source rows outside [0, T) zero, act = LeakyReLU(0.2) if lrelu.
 * x [B][T][Cin], y [B][T r][Cout] channels-last fp32; w2 = the r phases' weights [Cout][2 Cin] (tap 0 = W[:, :, p], tap 1 =
 * W[:, :, p + r]), each split (one scale) and fragment-packed (_lib.pack_conv_weights(.., taps = 2)), concatenated;
 * w_halves = r * 2 * Cout * 2 * Cin.  Cin % 32 == 0, Cout % 128 == 0. */
int ds_convt1d_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias, float* y, int B, int T,
                     int Cin, int Cout, int r, int pad, int lrelu, ds_stream_t stream);
/* MelGAN's two last upsampling layers in one pass each (vocoder/modules.py:104-113): y [B][2 Tin][Cout] =
 * ConvTranspose1d(k = 4, stride 2, padding 1)(LeakyReLU_0.2(x [B][Tin][Cin])), channels-last fp32.  w = the fp16 planes of the
 * polyphase weights * 2^s, [2 phases][Cout][2 taps][Cin] (phase p: W[:, :, p] on x[s0], W[:, :, p + 2] on x[s0 - 1]), w_plane
 * halves apart; out_scale = 2^-s.  (Cin, Cout) = (128, 64) and (64, 32) are built (ds_melgan_convt2_ok), an error elsewhere. */
int ds_melgan_convt2(const float* x, const void* w, long long w_plane, float out_scale, const float* bias, float* y, int B, int Tin,
                     int Cin, int Cout, ds_stream_t stream);
int ds_melgan_convt2_ok(int Cin, int Cout);
/* Generator tail (vocoder/modules.py:119-124) in one pass: out[b][t] = tanh(bias + sum_j sum_c w[j][c] *
 * LReLU_0.2(x[b][reflect(t + j - 3)][c])), x [B][T][C] channels-last fp32, w [7][C] fp32 (tap-major), out [B][T].
 * C = 32 (ngf = 32) is built; other widths use the 7-column GEMM + ds_stencil7_tanh. */
int ds_melgan_final(const float* x, const float* w, float bias, float* out, int B, int T, int C, ds_stream_t stream);
/* AttnBlock softmax (model.py:214-216): x[row][0..n) <- softmax(scale*x), x[row][n..ld) <- 0 */
int ds_softmax_rows(float* x, int rows, int n, int ld, float scale, ds_stream_t stream);
/* Encoder.conv_in (specvqgan/modules/diffusionmodules/model.py:423-427, :480): Conv2d(1, Cout, 3, stride 1, padding 1) on a
 * ONE-channel image x f32[B][H][W], w [Cout][9] (= weight [Cout][1][3][3]), bias [Cout] -> out f32[B][H][W][Cout]
 * (channels-last).  Direct fp32 multiply-adds, taps in the conv's order; store-bound.  Cout % 4 == 0, divides 1024.  gn_part
 * (may be null): [B][ds_conv3x3_c1_chunks(H, W)][2][Cout] doubles = per-channel sum / sum of squares of the output per row
 * (one chunk per image row), what ds_groupnorm_finish reads (the GroupNorm that follows needs no pass of its own over the
 * output).  W <= 2046. */
int ds_conv3x3_c1(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cout,
                  double* gn_part, ds_stream_t stream);
int ds_conv3x3_c1_chunks(int H, int W);
/* tap-sum for single-output-channel convs fed by a GEMM with N = taps */
int ds_stencil9(const float* taps, int ldt, float bias, float* out, int B, int H, int W, ds_stream_t stream);
int ds_stencil7_tanh(const float* taps, int ldt, float bias, float* out, int B, int N, ds_stream_t stream);
/* mel [B][C][T] -> [B][T][Cpad], y = a*x + b (generate_samples_batch.py:181-182 uses a = b = 0.5) */
int ds_mel_to_cl(const float* mel, float* out, int B, int C, int T, int Cpad, float a, float b,
                 ds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
