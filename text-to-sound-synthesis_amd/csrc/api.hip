// C ABI glue: error strings, the public gather-GEMM entry, and the native denoiser driver that
// enqueues the whole Text2ImageTransformer forward (transformer_utils.py:421-443) -- 19 blocks x 11
// launches -- from C++ on one HIP stream (no Python in the per-layer loop).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void ds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ds_version(void) { return 100; }
extern "C" const char* ds_last_error_string(void) { return g_err; }

static void fill(GemmParams& p, const ds_gemm_desc* d) {
    p.A = d->A; p.W = d->W; p.bias = d->bias; p.R = d->R; p.C = d->C;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldw = d->ldw > 0 ? d->ldw : d->K; p.ldc = d->ldc; p.ldr = d->ldr;
    p.groups = d->groups > 0 ? d->groups : 1;
    p.a_gstride = d->a_gstride; p.w_gstride = d->w_gstride; p.c_gstride = d->c_gstride;
    p.pro = d->pro; p.act = d->act; p.store = d->store; p.f16_round = d->f16_round;
    p.w3_plane = d->w3_plane;
    p.out_scale = d->out_scale;
    p.a_split = d->a_split; p.c_split = d->c_split; p.a_plane = d->a_plane; p.c_plane = d->c_plane;
    p.attn_kv = d->attn_kv; p.attn_heads = d->attn_heads; p.attn_nkey = d->attn_nkey; p.attn_qplane = d->attn_qplane;
    p.row_off = 0;
    p.pro_scale = d->pro_scale; p.pro_shift = d->pro_shift;
    p.rows_per_sample = d->rows_per_sample;
    p.Cin = d->Cin; p.H = d->H; p.W_ = d->Wd; p.up = d->up; p.taps = d->taps; p.dil = d->dil;
    p.ct_r = d->ct_r; p.ct_p = d->ct_p; p.ct_tin = d->ct_tin;
}

extern "C" int ds_gemm(const ds_gemm_desc* d, ds_stream_t stream) {
    DS_CHECK_ARG(d && d->A && d->W && d->C, "null pointer");
    DS_CHECK_ARG(d->pro == DS_PRO_NONE || d->pro == DS_PRO_LRELU || (d->pro_scale && d->pro_shift && d->Cin > 0),
                 "affine prologue needs pro_scale/pro_shift/Cin");
    DS_CHECK_ARG(d->store != DS_STORE_BATCH_T || d->rows_per_sample > 0, "BATCH_T store needs rows_per_sample");
    DS_CHECK_ARG(d->R == nullptr || d->store == DS_STORE_ROW, "residual only with row-major store");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    fill(p, d);
    return ds_launch_gemm(p, (hipStream_t)stream, d->loader);
}

// the conv-family loaders of the descriptor (DS_LOAD_CONV2D / CONV1D / CONVT1D / DENSE) on the fp16 matrix cores; W =
// split_f16x2 planes.  loader 0 (DS_LOAD_DENSE) in the descriptor of a caller that predates the other loaders meant the
// 3x3 conv: the conv2d geometry fields (H > 0 and K == 9 Cin) select it.
extern "C" int ds_conv2d_f16x2(const ds_gemm_desc* d, ds_stream_t stream) {
    DS_CHECK_ARG(d && d->A && d->W && d->C, "null pointer");
    DS_CHECK_ARG(d->w3_plane > 0, "w3_plane (plane stride of the split weights) is required");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    fill(p, d);
    const int loader = (d->loader == DS_LOAD_DENSE && d->H > 0 && d->Cin > 0 && d->K == 9 * d->Cin) ? DS_LOAD_CONV2D : d->loader;
    return ds_launch_conv2d_f16x2(p, (hipStream_t)stream, loader);
}

extern "C" int ds_gemm_f16x2(const ds_gemm_desc* d, ds_stream_t stream) {
    DS_CHECK_ARG(d && d->A && d->W && d->C, "null pointer");
    DS_CHECK_ARG(d->loader == DS_LOAD_DENSE && d->pro == DS_PRO_NONE && !d->f16_round,
                 "f16x2 is the dense, no-prologue kernel (groups: see ds_launch_gemm_f16x2)");
    DS_CHECK_ARG(d->w3_plane > 0, "w3_plane (plane stride of the split weights) is required");
    DS_CHECK_ARG(d->store != DS_STORE_BATCH_T || d->rows_per_sample > 0, "BATCH_T store needs rows_per_sample");
    DS_CHECK_ARG(d->R == nullptr || d->store == DS_STORE_ROW, "residual only with row-major store");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    fill(p, d);
    return ds_launch_gemm_f16x2(p, (hipStream_t)stream);
}

int ds_launch_gemm_f16x2_multi(const GemmParams* ps, int n, int cfg, hipStream_t stream);   // gemm_f16x2.hip
extern "C" int ds_gemm_f16x2_multi(const ds_gemm_desc* descs, int n, int cfg, ds_stream_t stream) {
    DS_CHECK_ARG(descs && n >= 1 && n <= 4, "1 .. 4 descriptors");
    GemmParams ps[4];
    memset(ps, 0, sizeof(ps));
    for (int i = 0; i < n; ++i) {
        const ds_gemm_desc* d = descs + i;
        DS_CHECK_ARG(d->A && d->W && d->C, "null pointer");
        DS_CHECK_ARG(d->loader == DS_LOAD_DENSE && d->pro == DS_PRO_NONE && !d->f16_round && d->w3_plane > 0,
                     "f16x2 is the dense, no-prologue kernel; w3_plane is required");
        fill(ps[i], d);
    }
    return ds_launch_gemm_f16x2_multi(ps, n, cfg, (hipStream_t)stream);
}

// ---- denoiser ---------------------------------------------------------------------------------------
struct ds_denoiser {
    ds_denoiser_desc d;
    std::vector<const float*> lp;  // [n_layer][DS_LP_COUNT]
    std::vector<const void*> lp3;  // split weights (same indexing), empty = fp32-MFMA mode
    std::vector<float> osc;        // f16x2: 2^-s per weight
    const void* w_logits3 = nullptr;
    float logits_osc = 1.f;
    int split_mode = DS_SPLIT_NONE;
    int pad_rows = 1;              // padded-row mode allowed (ds_denoiser_set_row_padding; see rows_per_sample())
    float S3(int layer, int slot) const { return osc.empty() ? 1.f : osc[(size_t)layer * DS_LP_COUNT + slot]; }
    const float* P(int layer, int slot) const { return lp[(size_t)layer * DS_LP_COUNT + slot]; }
    const void* P3(int layer, int slot) const { return lp3.empty() ? nullptr : lp3[(size_t)layer * DS_LP_COUNT + slot]; }
};

extern "C" int ds_denoiser_set_split_weights(ds_denoiser* h, int mode, const void* const* split3,
                                             const float* out_scales, const void* w_logits3, float logits_scale) {
    DS_CHECK_ARG(h, "null handle");
    if (mode == DS_SPLIT_NONE || !split3) {
        h->lp3.clear();
        h->osc.clear();
        h->w_logits3 = nullptr;
        h->split_mode = DS_SPLIT_NONE;
        return 0;
    }
    DS_CHECK_ARG(mode == DS_SPLIT_F16X2, "unknown split mode");
    DS_CHECK_ARG(out_scales && logits_scale > 0.f, "f16x2 needs the output scales");
    static const int need[6] = {DS_LP_W_QKV, DS_LP_W_PROJ1, DS_LP_W_Q2, DS_LP_W_PROJ2, DS_LP_W_FC1, DS_LP_W_FC2};
    for (int l = 0; l < h->d.n_layer; ++l)
        for (int s : need) DS_CHECK_ARG(split3[(size_t)l * DS_LP_COUNT + s], "missing split weight");
    DS_CHECK_ARG(w_logits3, "missing split logits weight");
    h->lp3.assign(split3, split3 + (size_t)h->d.n_layer * DS_LP_COUNT);
    h->osc.assign(out_scales, out_scales + (size_t)h->d.n_layer * DS_LP_COUNT);
    h->w_logits3 = w_logits3;
    h->logits_osc = logits_scale;
    h->split_mode = mode;
    return 0;
}

extern "C" int ds_denoiser_create(const ds_denoiser_desc* desc, const void* const* layer_ptrs, ds_denoiser** out) {
    DS_CHECK_ARG(desc && layer_ptrs && out, "null pointer");
    DS_CHECK_ARG(desc->n_embd == 1024 && desc->n_embd == desc->n_head * 64, "built for n_embd 1024, head dim 64");
    DS_CHECK_ARG(desc->n_codes == 256 || desc->n_codes == 512, "codebook size must be 256 or 512");
    DS_CHECK_ARG(desc->cond_dim % 32 == 0 && desc->cond_len <= 96 && desc->seq_len <= 288, "unsupported lengths");
    DS_CHECK_ARG(desc->tok_emb && desc->pos_emb && desc->lnf_g && desc->lnf_b && desc->w_logits && desc->b_logits &&
                     desc->sched,
                 "null weight pointer");
    for (int i = 0; i < desc->n_layer * DS_LP_COUNT; ++i) DS_CHECK_ARG(layer_ptrs[i], "null layer pointer");
    ds_denoiser* h = new ds_denoiser;
    h->d = *desc;
    h->lp.resize((size_t)desc->n_layer * DS_LP_COUNT);
    for (size_t i = 0; i < h->lp.size(); ++i) h->lp[i] = (const float*)layer_ptrs[i];
    *out = h;
    return 0;
}

extern "C" void ds_denoiser_destroy(ds_denoiser* h) { delete h; }

// workspace carve (floats): x, hn, qkv, att, fc, logits
struct Carve {
    float *x, *hn, *qkv, *att, *fc, *logits;
    float* kvimg;   // f16x2 mode: self-attention K / V^T images, inside the qkv region after the Q planes
};
// floats occupied by the K / V^T images of one attention over Lk keys (4 fp16 planes of nkey*64 per sample and head)
static size_t attn_img_floats(int B, int heads, int Lk) { return (size_t)B * heads * 4 * ds_attn_nkey(Lk) * 64 / 2; }
// Rows per sample of the activation matrices.  PADDED-ROW MODE (sampling steps in f16x2 mode at batch sizes whose GEMMs
// take the per-sample program, i.e. B = 64): every sample occupies PS_ROWS = 272 rows = 17 packed 16-row groups instead
// of L = 265, so a sample's tile starts on a packed group, and the tile's ninth block row is the 16 rows 256..271 on the
// 16x16x32 MFMA instead of 32 rows (gemm_f16x2_ps.hip NB16: -5 % MFMA work).  The 7 extra rows per sample start as zeros
// (ds_embed_rows), stay finite through every layer (LayerNorm of a constant row = its shift), are extra QUERIES of the
// attention (the ninth query tile exists anyway) but never keys (Lk = L masks them), and the sampler reads the L real
// rows of each sample (ds_sample_tail_rows).  Rows 256..264 of a sample are then summed in another order than by the
// 4-wave programs (same products): equal to ~1e-7 relative, not bit-identical across batch sizes.
static const int PS_ROWS = 272;
bool ds_gemm_f16x2_ps_taken(int B, int N);   // gemm_f16x2.hip: the dispatch rule of the per-sample program, forced tile included
static int rows_per_sample(const ds_denoiser* h, int B) {
    const int L = h->d.seq_len;
    if (!h->pad_rows || h->split_mode != DS_SPLIT_F16X2 || L > PS_ROWS || L <= PS_ROWS - 16) return L;
    return ds_gemm_f16x2_ps_taken(B, h->d.n_embd) ? PS_ROWS : L;    // the N = 1024 GEMMs' grid decides (gemm_f16x2.hip)
}
extern "C" int ds_denoiser_rows_per_sample(const ds_denoiser* h, int B) { return h && B > 0 ? rows_per_sample(h, B) : -1; }
extern "C" int ds_denoiser_set_row_padding(ds_denoiser* h, int on) {
    DS_CHECK_ARG(h, "null handle");
    h->pad_rows = on != 0;
    return 0;
}

static size_t carve(const ds_denoiser* h, int B, void* ws, Carve* c, int Lp = 0) {
    const size_t M = (size_t)B * (Lp > 0 ? Lp : PS_ROWS > h->d.seq_len ? PS_ROWS : h->d.seq_len), D = h->d.n_embd;
    const size_t M16 = (M + 15) & ~(size_t)15;   // hn / att / fc double as packed split planes (rows padded to 16)
    size_t off = 0;
    auto take = [&](size_t n) {
        float* p = ws ? (float*)ws + off : nullptr;
        off += (n + 63) & ~(size_t)63;
        return p;
    };
    Carve t;
    t.x = take(M * D);
    t.hn = take(M16 * D);
    // fp32 modes: [M][3D] rows.  f16x2 mode: Q planes (M*D floats) + the K / V^T images (padded to nkey key slots)
    const size_t qkv_ready = M * D + attn_img_floats(B, h->d.n_head, h->d.seq_len);   // (nkey covers 272 rows too)
    t.qkv = take(qkv_ready > M * 3 * D ? qkv_ready : M * 3 * D);
    t.kvimg = t.qkv ? t.qkv + M * D : nullptr;
    t.att = take(M16 * D);
    t.fc = take(M16 * D * h->d.mlp_mult);
    t.logits = take(M * h->d.n_codes);
    if (c) *c = t;
    return off * sizeof(float);
}

extern "C" int64_t ds_denoiser_workspace_bytes(const ds_denoiser* h, int B) {
    return h && B > 0 ? (int64_t)carve(h, B, nullptr, nullptr) : -1;
}
// caption K/V cache: fp32 modes [n_layer][B*cond_len][2D]; f16x2 mode [n_layer] K / V^T images + one layer of fp32
// rows as scratch for ds_denoiser_cond_kv.  The buffer is sized for either.
static size_t kv_layer_img_floats(const ds_denoiser* h, int B) { return attn_img_floats(B, h->d.n_head, h->d.cond_len); }
extern "C" int64_t ds_denoiser_kv_bytes(const ds_denoiser* h, int B) {
    if (!h || B <= 0) return -1;
    const size_t rows = (size_t)B * h->d.cond_len * 2 * h->d.n_embd;
    const size_t f32 = (size_t)h->d.n_layer * rows, img = (size_t)h->d.n_layer * kv_layer_img_floats(h, B) + rows;
    return (int64_t)((f32 > img ? f32 : img) * sizeof(float));
}

// ---- optional per-launch timing of the denoiser's GEMMs (bench.py's roofline leg) ---------------
struct ProfRec { hipEvent_t a, b; double flops; int tile; };
extern int g_last_tile;  // gemm_f32.hip
static bool g_prof = false;
static double g_prof_row_frac = 1.0;   // real rows / launched rows of the GEMMs being recorded (padded-row mode: 265 / 272)
static std::vector<ProfRec> g_recs;

extern "C" int ds_profile_enable(int on) {
    g_prof = on != 0;
    return 0;
}
// Waits for the recorded launches; per GEMM program c (0: 128x128, 1: 128x64, 2: 64x64 tiles, 3: the per-sample
// ping-pong program on full tiles, 4: on half tiles -- each is its own kernel symbol) returns summed duration ms[c],
// algorithmic flops[c] (2MNK) and launches[c] for the first n <= 5 programs; the four-program form folds 4 into 3.
extern "C" int ds_profile_collect_n(double* ms, double* flops, int64_t* launches, int n) {
    DS_CHECK_ARG(ms && flops && launches && n >= 4 && n <= 5, "bad arguments");
    for (int c = 0; c < n; ++c) { ms[c] = 0.0; flops[c] = 0.0; launches[c] = 0; }
    for (auto& r : g_recs) {
        const int c = r.tile < n ? r.tile : n - 1;
        float e = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&e, r.a, r.b) == hipSuccess) ms[c] += e;
        flops[c] += r.flops;
        launches[c] += 1;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    return 0;
}
extern "C" int ds_profile_collect(double* ms, double* flops, int64_t* launches) {
    return ds_profile_collect_n(ms, flops, launches, 4);
}

static int dense(const float* A, int lda, const float* W, const float* bias, const float* R, float* C, int ldc,
                 int M, int N, int K, int act, hipStream_t s, int store = DS_STORE_ROW, int rps = 0,
                 const void* W3 = nullptr, int split_mode = DS_SPLIT_NONE, float osc = 1.f,
                 long long a_plane = 0, long long c_plane = 0, void* attn_kv = nullptr, int attn_heads = 0,
                 int attn_nkey = 0, long long attn_qplane = 0, long long w_plane = 0) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.R = R; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = K; p.ldc = ldc; p.ldr = ldc;
    p.groups = 1; p.act = act; p.store = store; p.rows_per_sample = rps;
    if (W3) {  // fp32-class GEMM on the fp16 matrix cores
        p.W = (const float*)W3;
        p.w3_plane = (long long)N * K;
        p.out_scale = osc;
        p.a_split = a_plane > 0; p.a_plane = a_plane;   // f16x2 only: A and W are packed split planes
        if (a_plane > 0) p.w3_plane = (long long)((N + 15) & ~15) * K;
        p.c_split = c_plane > 0; p.c_plane = c_plane;
        p.attn_kv = attn_kv; p.attn_heads = attn_heads; p.attn_nkey = attn_nkey; p.attn_qplane = attn_qplane;
        if (w_plane > 0) p.w3_plane = w_plane;   // a row range of a larger packed weight keeps that weight's stride
    }
    auto launch = [&]() {
        if (!W3) return ds_launch_gemm(p, s, DS_LOAD_DENSE);
        return ds_launch_gemm_f16x2(p, s);
    };
    if (!g_prof) return launch();
    ProfRec r;
    r.flops = 2.0 * M * N * K * g_prof_row_frac;   // algorithmic: the padding rows of the padded-row mode do not count
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
        ds_set_error("profile: hipEventCreate failed");
        return -2;
    }
    (void)hipEventRecord(r.a, s);
    const int rc = launch();
    (void)hipEventRecord(r.b, s);
    r.tile = g_last_tile;
    g_recs.push_back(r);
    return rc;
}

#define TRY(x)            \
    do {                  \
        int rc_ = (x);    \
        if (rc_) return rc_; \
    } while (0)

extern "C" int ds_denoiser_cond_kv(const ds_denoiser* h, const float* cond, int B, float* kv, ds_stream_t stream) {
    DS_CHECK_ARG(h && cond && kv && B > 0, "bad arguments");
    const int D = h->d.n_embd, Mc = B * h->d.cond_len;
    if (h->split_mode == DS_SPLIT_F16X2) {   // attention-ready K / V^T images per layer (see ds_denoiser_kv_bytes)
        const size_t per = kv_layer_img_floats(h, B);
        float* rows = kv + (size_t)h->d.n_layer * per;
        for (int l = 0; l < h->d.n_layer; ++l) {
            TRY(dense(cond, h->d.cond_dim, h->P(l, DS_LP_W_KV2), h->P(l, DS_LP_B_KV2), nullptr, rows, 2 * D, Mc, 2 * D,
                      h->d.cond_dim, DS_ACT_NONE, (hipStream_t)stream));
            TRY(ds_attn_pack_kv(rows, 2 * D, D, kv + (size_t)l * per, B, h->d.n_head, h->d.cond_len, stream));
        }
        return 0;
    }
    for (int l = 0; l < h->d.n_layer; ++l)
        TRY(dense(cond, h->d.cond_dim, h->P(l, DS_LP_W_KV2), h->P(l, DS_LP_B_KV2), nullptr,
                  kv + (size_t)l * Mc * 2 * D, 2 * D, Mc, 2 * D, h->d.cond_dim, DS_ACT_NONE, (hipStream_t)stream));
    return 0;
}

// Lp: rows per sample of every activation matrix (d.seq_len, or PS_ROWS in padded-row mode: layout 0 only)
static int forward_impl(const ds_denoiser* h, const int64_t* tokens, const int64_t* t, const float* kv, int B,
                        const Carve& w, float* logits, int layout, hipStream_t s, int Lp, bool zero_kv_pad = true) {
    const ds_denoiser_desc& d = h->d;
    const int D = d.n_embd, Lv = d.seq_len, L = Lp, M = B * L, Mc = B * d.cond_len, F = D * d.mlp_mult;
    DS_CHECK_ARG(Lp == Lv || (layout == 0 && Lp > Lv), "padded rows need the row-major logits layout");
    g_prof_row_frac = (double)Lv / Lp;
    const float scale = 0.125f;  // 1/sqrt(64), transformer_utils.py:48
    // f16x2 mode: attention follows the GEMM arithmetic, and every GEMM input is produced as PACKED SPLIT PLANES
    // (common.h ds_packed_off; two fp16 planes in the bytes of the fp32 buffer, rows padded to 16) by the kernel
    // before it, with the same ds_split_hi/lo code the GEMM loader would run -- results are bit-identical to
    // splitting in the loader, and the GEMM stages its tiles by LDS-DMA.
    const bool f16 = h->split_mode == DS_SPLIT_F16X2;
    const long long M16 = (M + 15) & ~15;
    const long long pD = f16 ? M16 * D : 0, pF = f16 ? M16 * F : 0;
    auto adaln = [&](int l, int slot) {
        return f16 ? ds_adaln_split(w.x, w.hn, M, L, D, h->P(l, slot), t, s)
                   : ds_adaln(w.x, w.hn, M, L, D, h->P(l, slot), t, s);
    };
    auto lnorm = [&](const float* g, const float* b_) {
        return f16 ? ds_layernorm_split(w.x, w.hn, M, D, g, b_, s) : ds_layernorm(w.x, w.hn, M, D, g, b_, s);
    };
    auto attn = [&](const float* q, int ldq, const float* k, const float* v, int ldkv, int Lk) {
        return ds_attention(q, ldq, k, ldkv, v, ldkv, w.att, D, B, d.n_head, L, Lk, scale, s);
    };
    // f16x2 mode: the QKV / cross-Q GEMMs write attention-ready operands (Q planes at w.qkv, K / V^T images at
    // w.kvimg; the caption images come from ds_denoiser_cond_kv) and the attention kernel stages them by LDS-DMA
    const long long qpl = (long long)M * D;   // halves per Q plane ([B][heads][L][64])
    const int nkS = ds_attn_nkey(L);
    auto lin_attn = [&](int l, int slot, int bslot, int N, int n_weight, void* imgs) {
        return dense(w.hn, D, h->P(l, slot), h->P(l, bslot), nullptr, w.qkv, N, M, N, D, DS_ACT_NONE, s, DS_STORE_ATTN, L,
                     h->P3(l, slot), h->split_mode, h->S3(l, slot), pD, 0, imgs, d.n_head, nkS, qpl,
                     (long long)n_weight * D);
    };
    auto attn_ready = [&](const void* imgs, int Lk) {
        return ds_attention_f16x2_ready(w.qkv, qpl, imgs, w.att, D, B, d.n_head, L, Lk, scale, s);
    };
    // key slots L..nkey-1 of the self-attention images are never written: they must read as zero (a chain of steps over one
    // workspace zeroes them in its first step only: nothing between two of its steps touches the workspace)
    if (f16 && zero_kv_pad) {
        hipError_t e = hipMemsetAsync(w.kvimg, 0, attn_img_floats(B, d.n_head, L) * sizeof(float), s);
        if (e != hipSuccess) {
            ds_set_error("forward: hipMemsetAsync: %s", hipGetErrorString(e));
            return -2;
        }
    }
    // y = act(A W^T + b) (+ R) for layer-l weight `slot`; A (and optionally C) pre-split in f16x2 mode
    auto lin = [&](int l, int slot, int bslot, const float* A, int lda, long long ap, const float* R, float* C, int N,
                   int K, int act, long long cp) {
        return dense(A, lda, h->P(l, slot), h->P(l, bslot), R, C, N, M, N, K, act, s, DS_STORE_ROW, L, h->P3(l, slot),
                     h->split_mode, h->S3(l, slot), ap, cp);
    };
    TRY(ds_embed_rows(tokens, d.tok_emb, d.pos_emb, w.x, B, Lv, L, D, s));
    for (int l = 0; l < d.n_layer; ++l) {
        // x += attn1(ln1(x, t))
        TRY(adaln(l, DS_LP_ADALN1));
        if (f16) {
            TRY(lin_attn(l, DS_LP_W_QKV, DS_LP_B_QKV, 3 * D, 3 * D, w.kvimg));   // Q planes + K / V^T images
            TRY(attn_ready(w.kvimg, Lv));                                          // padded rows are queries, never keys
        } else {
            TRY(lin(l, DS_LP_W_QKV, DS_LP_B_QKV, w.hn, D, pD, nullptr, w.qkv, 3 * D, D, DS_ACT_NONE, 0));
            TRY(attn(w.qkv, 3 * D, w.qkv + D, w.qkv + 2 * D, 3 * D, Lv));
        }
        TRY(lin(l, DS_LP_W_PROJ1, DS_LP_B_PROJ1, w.att, D, pD, w.x, w.x, D, D, DS_ACT_NONE, 0));
        // x += attn2(ln1_1(x, t), cond)
        TRY(adaln(l, DS_LP_ADALN2));
        if (f16) {
            TRY(lin_attn(l, DS_LP_W_Q2, DS_LP_B_Q2, D, D, nullptr));
            TRY(attn_ready(kv + (size_t)l * kv_layer_img_floats(h, B), d.cond_len));
        } else {
            TRY(lin(l, DS_LP_W_Q2, DS_LP_B_Q2, w.hn, D, pD, nullptr, w.qkv, D, D, DS_ACT_NONE, 0));
            const float* kvl = kv + (size_t)l * Mc * 2 * D;
            TRY(attn(w.qkv, D, kvl, kvl + D, 2 * D, d.cond_len));
        }
        TRY(lin(l, DS_LP_W_PROJ2, DS_LP_B_PROJ2, w.att, D, pD, w.x, w.x, D, D, DS_ACT_NONE, 0));
        // x += mlp(ln2(x))
        TRY(lnorm(h->P(l, DS_LP_LN2_G), h->P(l, DS_LP_LN2_B)));
        TRY(lin(l, DS_LP_W_FC1, DS_LP_B_FC1, w.hn, D, pD, nullptr, w.fc, F, D, DS_ACT_GELU2, pF));
        TRY(lin(l, DS_LP_W_FC2, DS_LP_B_FC2, w.fc, F, pF, w.x, w.x, D, F, DS_ACT_NONE, 0));
    }
    TRY(lnorm(d.lnf_g, d.lnf_b));
    if (layout == 0)
        TRY(dense(w.hn, D, d.w_logits, d.b_logits, nullptr, logits, d.n_codes, M, d.n_codes, D, DS_ACT_NONE, s,
                  DS_STORE_ROW, 0, h->w_logits3, h->split_mode, h->logits_osc, pD));
    else
        TRY(dense(w.hn, D, d.w_logits, d.b_logits, nullptr, logits, L, M, d.n_codes, D, DS_ACT_NONE, s,
                  DS_STORE_BATCH_T, L, h->w_logits3, h->split_mode, h->logits_osc, pD));
    return 0;
}

extern "C" int ds_denoiser_forward(const ds_denoiser* h, const int64_t* tokens, const int64_t* t, const float* kv,
                                   int B, void* workspace, float* logits, int logits_layout, ds_stream_t stream) {
    DS_CHECK_ARG(h && tokens && t && kv && workspace && logits && B > 0, "bad arguments");
    Carve w;
    carve(h, B, workspace, &w, h->d.seq_len);
    return forward_impl(h, tokens, t, kv, B, w, logits, logits_layout, (hipStream_t)stream, h->d.seq_len);
}

// t drives the network (AdaLN), t_post the posterior: they differ only for the skip-step sampler
// (sample_fast, diffusion_transformer.py:796-803 calls q_posterior with t - skip_step)
static int step_impl(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t, const int64_t* t_post,
                     const float* kv, const float* u, const int64_t* gids, unsigned long long seed, int call, int B,
                     int initial, float trunc_r, int trunc_k, void* workspace, int64_t* tokens_out, ds_stream_t stream,
                     bool zero_kv_pad = true) {
    Carve w;
    const int Lp = rows_per_sample(h, B);
    carve(h, B, workspace, &w, Lp);
    TRY(forward_impl(h, tokens_in, t, kv, B, w, w.logits, 0, (hipStream_t)stream, Lp, zero_kv_pad));
    return ds_sample_tail_rows(w.logits, Lp, tokens_in, t_post ? t_post : t, u, h->d.sched, tokens_out, nullptr, nullptr,
                               nullptr, B, h->d.seq_len, h->d.n_codes, h->d.n_steps, initial, trunc_r, trunc_k, stream,
                               gids, seed, call);
}

extern "C" int ds_denoiser_step_ex(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t,
                                   const int64_t* t_post, const float* kv, const float* u, int B, int initial,
                                   float trunc_r, int trunc_k, void* workspace, int64_t* tokens_out,
                                   ds_stream_t stream) {
    DS_CHECK_ARG(h && tokens_in && t && kv && u && workspace && tokens_out && B > 0, "bad arguments");
    return step_impl(h, tokens_in, t, t_post, kv, u, nullptr, 0ull, 0, B, initial, trunc_r, trunc_k, workspace, tokens_out,
                     stream);
}

// the same step with the noise drawn inside the sampler kernel (sampler.hip: Philox keyed by seed, counter = global
// caption id gids[b], sampler call index, grid position, class): what a caption draws does not depend on the batch it is
// in, on its position in it, or on the rank that runs it
extern "C" int ds_denoiser_step_rng(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t,
                                    const int64_t* t_post, const float* kv, const int64_t* gids,
                                    unsigned long long seed, int call, int B, int initial, float trunc_r, int trunc_k,
                                    void* workspace, int64_t* tokens_out, ds_stream_t stream) {
    DS_CHECK_ARG(h && tokens_in && t && kv && gids && workspace && tokens_out && B > 0, "bad arguments");
    return step_impl(h, tokens_in, t, t_post, kv, nullptr, gids, seed, call, B, initial, trunc_r, trunc_k, workspace,
                     tokens_out, stream);
}

// A whole reverse chain enqueued from C++ (DiffusionTransformer.sample's loop, diffusion_transformer.py:639-641, and
// sample_fast's, :790-804): n_calls steps; t_steps is a DEVICE array i64[n_calls][2][B] -- per call the network's
// timestep vector and the posterior's (equal except for the skip-step sampler).  tokens (in: the start state, out: the
// result) and tokens_tmp ([B][seq_len] each) are the two ends of the ping-pong; call k uses Philox call index
// call0 + k; `initial` marks the first call's state as the all-[MASK] start.  Nothing returns to the host between steps.
extern "C" int ds_denoiser_sample_rng(const ds_denoiser* h, int64_t* tokens, int64_t* tokens_tmp, const int64_t* t_steps,
                                      int n_calls, const float* kv, const int64_t* gids, unsigned long long seed,
                                      int call0, int B, int initial, float trunc_r, int trunc_k, void* workspace,
                                      ds_stream_t stream) {
    DS_CHECK_ARG(h && tokens && tokens_tmp && t_steps && kv && gids && workspace && B > 0 && n_calls >= 0, "bad arguments");
    int64_t *cur = tokens, *nxt = tokens_tmp;
    for (int k = 0; k < n_calls; ++k) {
        const int64_t* tk = t_steps + (size_t)k * 2 * B;
        TRY(step_impl(h, cur, tk, tk + B, kv, nullptr, gids, seed, call0 + k, B, initial && k == 0, trunc_r, trunc_k,
                      workspace, nxt, stream, k == 0));
        int64_t* sw = cur; cur = nxt; nxt = sw;
    }
    if (cur != tokens) {
        hipError_t e = hipMemcpyAsync(tokens, cur, (size_t)B * h->d.seq_len * sizeof(int64_t), hipMemcpyDeviceToDevice,
                                      (hipStream_t)stream);
        if (e != hipSuccess) {
            ds_set_error("ds_denoiser_sample_rng: hipMemcpyAsync: %s", hipGetErrorString(e));
            return -2;
        }
    }
    return 0;
}

extern "C" int ds_denoiser_step(const ds_denoiser* h, const int64_t* tokens_in, const int64_t* t, const float* kv,
                                const float* u, int B, int initial, float trunc_r, void* workspace,
                                int64_t* tokens_out, ds_stream_t stream) {
    return ds_denoiser_step_ex(h, tokens_in, t, nullptr, kv, u, B, initial, trunc_r, 0, workspace, tokens_out, stream);
}
