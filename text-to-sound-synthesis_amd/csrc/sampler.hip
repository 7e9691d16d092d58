// The per-column tail of one reverse-diffusion step, fused into one kernel (one wavefront per
// grid position (b, pos); K codes = NPL*64 so each lane owns NPL classes; [MASK] is class K):
//
//   predict_start      diffusion_transformer.py:285-289   log_softmax in float64 -> f32, -70 row, clamp
//   top-r truncation   models/dalle_spec.py:158-174       keep class iff mass ranked before it < r
//   q_posterior        diffusion_transformer.py:293-339   (+ q_pred :253-267, q_pred_one_timestep
//                                                          :241-251, log_add_exp :28-30)
//   log_sample_categorical  :359-368                      Gumbel-argmax with the caller's uniforms
//
// State between steps is the token index, not the reference's 272 KB/sample log-one-hot: the
// log-one-hot is only ever consumed through argmax / q_pred / q_pred_one_timestep, which need
// {0 at x_t, log(1e-30) elsewhere}; the all-[MASK] start state is {0, -inf} (:633-636) and is
// flagged by `initial`.
//
// The reference sorts each column (2 sorts + cumsum + gather); here the "mass ranked before me"
// is an O(K^2/64) compare-and-add per lane against an LDS copy of the column -- 66k flops per
// column, nothing next to the 155 GFLOP transformer step, and no sort network.
// Noise is read in the reference's own [B, K+1, L] layout so that torch.rand_like on the same
// shape reproduces the reference's RNG stream -- or (the *_rng entries) drawn inside the kernel from a counter-based
// Philox4x32-10 stream keyed by (seed; global caption id, sampler call, grid position, class): the draw of a caption
// then depends neither on the batch it is in, nor on its position in it, nor on the rank that runs it (SURVEY.md
// section 8e: "generate the noise per sample, seeded by global caption index"), no [B][K+1][L] tensor of uniforms
// exists, and a whole reverse chain can be enqueued without returning to the host (api.hip ds_denoiser_sample_rng).
#include "common.h"

// ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 constants) ----
// counter (c0, c1, c2, c3), key (k0, k1) -> four 32-bit words.  Host mirror: text_to_sound_synthesis_amd/shard.py
// philox4x32_10 / caption_uniforms (numpy), checked against the published known-answer vectors in tests/.
typedef unsigned ds_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ds_u32x4 ds_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                      unsigned k1) {
    asm volatile("" : "+v"(k0), "+v"(k1), "+v"(c1), "+v"(c2), "+v"(c3));   // everything in VGPRs: scalar rounds push the sampler kernel past its SGPRs
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return ds_u32x4{c0, c1, c2, c3};
}
// The uniform of class c = 64 j + lane (j = c >> 6; the [MASK] class K is j = K / 64, lane 0) at grid position pos of
// caption gid in sampler call `call` of stream `sid` (0: reverse steps, 1: q_sample) is word (j & 3) of
//   Philox(counter = (64 (j >> 2) + lane, pos | sid << 16, call, gid), key = seed)  ->  (word >> 8) * 2^-24  in [0, 1):
// the lane that owns classes 64 j + lane runs one Philox per four of them and uses every word.
__device__ __forceinline__ float ds_u01(unsigned w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }

#define LOG_ZERO_F (-69.07755278982137f)  // logf(1e-30f)

__device__ __forceinline__ float lae(float a, float b) {  // log(exp a + exp b)
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float wmaxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wsumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wsumd(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct SampleParams {
    const float* logits;      // [B*L][K]  (row-major, class contiguous)
    const int64_t* xt;        // [B][L] current tokens
    const int64_t* t;         // [B]
    const float* u;           // [B][K+1][L] uniforms in [0,1)
    const float* sched;       // 8 rows of (T+1) floats: log_at, log_bt, log_ct, log_1_min_ct,
                              //   log_cumprod_at, log_cumprod_bt, log_cumprod_ct, log_1_min_cumprod_ct
    int64_t* out_tokens;      // [B][L]
    float* dbg_log_pred;      // optional [B][K+1][L]
    float* dbg_trunc;         // optional
    float* dbg_post;          // optional
    int B, L, T;
    int initial;              // 1: x_t is the all-[MASK] start state (log one-hot = 0 / -inf)
    float trunc_r;            // < 0: no top-r truncation
    int trunc_k;              // > 0: top-k truncation instead ('top{k}p', dalle_spec.py:147-157)
    int lrows;                // rows of `logits` per sample (>= L: the denoiser's padded-row mode), L by default
};
// u == nullptr: the uniforms come from the Philox stream of (seed; gid[b], call) instead (see ds_u01 above).  (A separate
// kernel argument: with these fields inside SampleParams hipcc reserves 68 bytes of -- unused -- private segment.)
struct SampleRng {
    const int64_t* gid;       // [B] global caption ids (< 2^32)
    unsigned seed_lo, seed_hi;
    int call;
};

template <int NPL, bool RNG>
__global__ __launch_bounds__(256) void ds_sample_tail_kernel(const SampleParams p, const SampleRng g) {
    constexpr int K = NPL * 64;
    __shared__ float s_lp[4][K];
    __shared__ __attribute__((aligned(16))) float s_pr[4][K];
    __shared__ unsigned long long s_key[4][K];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int col = blockIdx.x * 4 + w;
    const bool live = col < p.B * p.L;  // a dead wave shadows the last column and writes nothing
    if (!live) col = p.B * p.L - 1;
    const int b = col / p.L, pos = col - b * p.L;

    // ---- predict_start: float64 log-softmax over the K real classes ----
    float v[NPL];
    const float* lg = p.logits + ((size_t)b * p.lrows + pos) * K;
#pragma unroll
    for (int j = 0; j < NPL; ++j) v[j] = lg[j * 64 + lane];
    float mx = v[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, v[j]);
    mx = wmaxf(mx);
    double se = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) se += exp((double)v[j] - (double)mx);
    se = wsumd(se);
    const double lse64 = log(se);
    float lp[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        float x = (float)(((double)v[j] - (double)mx) - lse64);
        lp[j] = fminf(fmaxf(x, -70.f), 0.f);
    }
    const size_t dbg_base = (size_t)b * (K + 1) * p.L + pos;
    if (p.dbg_log_pred && live) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) p.dbg_log_pred[dbg_base + (size_t)(j * 64 + lane) * p.L] = lp[j];
        if (lane == 0) p.dbg_log_pred[dbg_base + (size_t)K * p.L] = -70.f;
    }

    // ---- truncation: top-r (probability mass ranked ahead < r) or top-k (rank < k) ----
    // top-k, dalle_spec.py:147-157: topk over the K+1 rows, everything else -70.  The [MASK] row is -70 and a
    // kept -70 is indistinguishable from a dropped one, so ranking the K real classes is equivalent.
    float tr[NPL];
    if (p.trunc_k > 0) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) s_lp[w][j * 64 + lane] = lp[j];
        __syncthreads();
        int rank[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) rank[j] = 0;
        for (int c = 0; c < K; ++c) {
            const float ol = s_lp[w][c];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int me = j * 64 + lane;
                rank[j] += (ol > lp[j] || (ol == lp[j] && c < me)) ? 1 : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) tr[j] = rank[j] < p.trunc_k ? lp[j] : -70.f;
    } else if (p.trunc_r >= 0.f) {
        // dalle_spec.py:159-172: sort descending, exp, cumsum; rank i is kept iff the mass of ranks 0 .. i-1 is < r (rank 0 always).
        // Round 4: the ranks come from a bitonic sort of the column in LDS (one wave per column, 64-bit keys = order-preserving
        // image of the log-probability | inverted class index: descending value, ascending index among ties) and the mass from
        // the sum in rank order, accumulated the way torch's CPU cumsum does, instead of 65 536 pair tests per column
        // (the kernel's 173 us per step were 160 us of those: the largest item of a sampling step after the transformer).
        // The kept set is a prefix of the rank order (the masses are non-negative), so the scan only has to find its length.
        unsigned long long* key = s_key[w];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const unsigned bits = __float_as_uint(lp[j] + 0.f);                         // (+ 0: -0 and +0 are one key)
            const unsigned u = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);      // unsigned order = float order
            key[j * 64 + lane] = ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)(j * 64 + lane));
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = 2; k <= K; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                for (int h = 0; h < K / 128; ++h) {
                    const int tq = lane + 64 * h;
                    const int i = ((tq & ~(jj - 1)) << 1) | (tq & (jj - 1)), l = i | jj;
                    const unsigned long long a = key[i], bq = key[l];
                    const bool sw = ((i & k) == 0) ? (a < bq) : (a > bq);                // descending blocks where (i & k) == 0
                    if (sw) { key[i] = bq; key[l] = a; }
                }
                __builtin_amdgcn_wave_barrier();       // (a wave's LDS operations execute in order: no hardware barrier needed)
            }
        // probabilities in rank order (this lane: ranks 4 lane' .. for the scan's 16-byte reads)
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int rk = j * 64 + lane;
            const unsigned u = (unsigned)(key[rk] >> 32);
            const unsigned bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            s_pr[w][rk] = expf(__uint_as_float(bits));
        }
        __builtin_amdgcn_wave_barrier();
        // n_keep = 1 + #{ i >= 1 : p[0] + .. + p[i-1] < r }, every lane runs the same sequential sum (uniform early exit)
        int n_keep = 1;
        {
            // torch's CPU cumsum of a float tensor accumulates in DOUBLE and rounds every partial sum to float: the same here
            // (tests/test_sampler_sort_emulation.py: with this accumulator the restated algorithm equals the reference's
            // truncation AS EXECUTED ON THE CPU bit for bit -- the goldens are CPU runs; a plain fp32 chain differs in the
            // last place of some partial sums, and so does a CUDA execution of dalle_spec.py:163-165, whose cumsum is a
            // parallel fp32 scan: against a GPU run of the reference a cut at a rounding-level near-tie can differ)
            // float(c) < r, tested on the double: float(c) <= pf = the float below r  <=>  c below the midpoint of pf and r (at the
            // midpoint itself round-to-nearest-even goes to pf iff pf's last mantissa bit is 0)
            double cum = 0.0;
            const float r_ = p.trunc_r;
            const float pf = __uint_as_float(__float_as_uint(r_) - 1u);      // r > 0 (top-r rates are in (0, 1])
            const double mid = 0.5 * ((double)pf + (double)r_);
            const bool tie_low = (__float_as_uint(pf) & 1u) == 0u;
            // c <= mid as ONE comparison: c < the double after mid (mid > 0: its bit pattern + 1)
            const double lim = tie_low ? __longlong_as_double(__double_as_longlong(mid) + 1) : mid;
            auto below = [&](double c) { return c < lim; };
            if (r_ > 0.f)            // (r = 0: no partial sum is below it, rank 0 alone survives)
            for (int i = 0; i < K; i += 4) {
                const f32x4 p4 = *(const f32x4*)(&s_pr[w][i]);
                const double c0 = cum + (double)p4[0], c1 = c0 + (double)p4[1], c2 = c1 + (double)p4[2], c3 = c2 + (double)p4[3];
                // rank i + e + 1 is kept iff float(c_e) < r; the sums do not decrease, so the tests fail from some e on
                const bool b3 = below(c3);
                n_keep += (below(c0) ? 1 : 0) + (below(c1) ? 1 : 0) + (below(c2) ? 1 : 0) + ((b3 && i + 4 < K) ? 1 : 0);
                cum = c3;
                if (!b3) break;
            }
        }
        // back to class order: the class at rank rk is kept iff rk < n_keep
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int rk = j * 64 + lane;
            const unsigned cls = 0xffffffffu - (unsigned)(key[rk] & 0xffffffffull);
            s_lp[w][cls] = rk < n_keep ? 1.f : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NPL; ++j) tr[j] = s_lp[w][j * 64 + lane] != 0.f ? lp[j] : -70.f;
    } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j) tr[j] = lp[j];
    }
    if (p.dbg_trunc && live) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) p.dbg_trunc[dbg_base + (size_t)(j * 64 + lane) * p.L] = tr[j];
        if (lane == 0) p.dbg_trunc[dbg_base + (size_t)K * p.L] = -70.f;
    }

    // ---- q_posterior ----
    const int T1 = p.T + 1;
    const int t = (int)p.t[b];
    const int tm1 = (t - 1 + T1) % T1;
    const float* S = p.sched;
    const float lat = S[0 * T1 + t], lbt = S[1 * T1 + t], lct = S[2 * T1 + t];
    const float lcat = S[4 * T1 + t], lcbt = S[5 * T1 + t], lcct = S[6 * T1 + t];
    const float lcat1 = S[4 * T1 + tm1], lcbt1 = S[5 * T1 + tm1], lcct1 = S[6 * T1 + tm1], l1mcct1 = S[7 * T1 + tm1];
    const int xt = (int)p.xt[col];
    const bool is_mask = xt == K;
    const float off = p.initial ? -INFINITY : LOG_ZERO_F;  // log one-hot value away from x_t
    // log q(x_t | x_0 = c) and log q(x_t | x_{t-1} = c) take two values per column: c == x_t or not
    const float qt_hit = is_mask ? lcct : lae(0.f + lcat, lcbt);
    const float qt_off = is_mask ? lcct : lae(off + lcat, lcbt);
    const float q1_hit = is_mask ? lct : lae(0.f + lat, lbt);
    const float q1_off = is_mask ? lct : lae(off + lat, lbt);
    const float qt_m = is_mask ? 0.f : LOG_ZERO_F;  // [MASK] row
    const float q1_m = qt_m;

    float q[NPL];
    float qmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = j * 64 + lane;
        q[j] = tr[j] - (c == xt ? qt_hit : qt_off);
        qmax = fmaxf(qmax, q[j]);
    }
    const float q_m = -70.f - qt_m;
    qmax = fmaxf(wmaxf(qmax), q_m);
    float es = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) es += expf(q[j] - qmax);
    es = wsumf(es) + expf(q_m - qmax);
    const float lse = logf(es) + qmax;  // torch.logsumexp

    float post[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = j * 64 + lane;
        const float ev = lae((q[j] - lse) + lcat1, lcbt1);
        const float o = ev + (c == xt ? q1_hit : q1_off) + lse;
        post[j] = fminf(fmaxf(o, -70.f), 0.f);
    }
    float post_m;
    {
        const float ev = lae((q_m - lse) + l1mcct1, lcct1);
        post_m = fminf(fmaxf(ev + q1_m + lse, -70.f), 0.f);
    }
    if (p.dbg_post && live) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) p.dbg_post[dbg_base + (size_t)(j * 64 + lane) * p.L] = post[j];
        if (lane == 0) p.dbg_post[dbg_base + (size_t)K * p.L] = post_m;
    }

    // ---- Gumbel-argmax (first index wins ties, as torch.argmax) ----
    float un[NPL], un_m;            // this lane's uniforms (classes 64 j + lane) and the [MASK] class's
    if (!RNG) {
        const float* up = p.u + dbg_base;
#pragma unroll
        for (int j = 0; j < NPL; ++j) un[j] = up[(size_t)(j * 64 + lane) * p.L];
        un_m = up[(size_t)K * p.L];
    } else {
        const unsigned gid = (unsigned)g.gid[b];
#pragma unroll
        for (int q = 0; q < NPL / 4; ++q) {
            const ds_u32x4 w4 = ds_philox4x32_10(q * 64 + lane, pos, g.call, gid, g.seed_lo, g.seed_hi);
            un[4 * q + 0] = ds_u01(w4.x); un[4 * q + 1] = ds_u01(w4.y);
            un[4 * q + 2] = ds_u01(w4.z); un[4 * q + 3] = ds_u01(w4.w);
        }
        un_m = ds_u01(ds_philox4x32_10((NPL / 4) * 64, pos, g.call, gid, g.seed_lo, g.seed_hi).x);
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = j * 64 + lane;
        const float gsc = -logf(-logf(un[j] + 1e-30f) + 1e-30f) + post[j];
        if (gsc > best) { best = gsc; bidx = c; }  // ascending c within a lane keeps the first max
    }
    if (lane == 0) {
        const float gsc = -logf(-logf(un_m + 1e-30f) + 1e-30f) + post_m;
        if (gsc > best) { best = gsc; bidx = K; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0 && live) p.out_tokens[col] = bidx;
}

// ---- training-loss terms (DiffusionTransformer._train_loss, diffusion_transformer.py:408-476), forward only ----
// q_posterior (:293-339) of a per-column distribution lx0[] over the K classes (+ its [MASK] row lx0_m) given
// x_t and t: the same arithmetic as the sampling tail above, factored out for the two posteriors of the loss.
template <int NPL>
__device__ __forceinline__ void ds_posterior(const float (&lx0)[NPL], float lx0_m, int xt, int K, int lane,
                                             const float* __restrict__ S, int T1, int t, float (&post)[NPL],
                                             float& post_m) {
    const int tm1 = (t - 1 + T1) % T1;
    const float lat = S[0 * T1 + t], lbt = S[1 * T1 + t], lct = S[2 * T1 + t];
    const float lcat = S[4 * T1 + t], lcbt = S[5 * T1 + t], lcct = S[6 * T1 + t];
    const float lcat1 = S[4 * T1 + tm1], lcbt1 = S[5 * T1 + tm1], lcct1 = S[6 * T1 + tm1], l1mcct1 = S[7 * T1 + tm1];
    const bool is_mask = xt == K;
    const float qt_hit = is_mask ? lcct : lae(0.f + lcat, lcbt), qt_off = is_mask ? lcct : lae(LOG_ZERO_F + lcat, lcbt);
    const float q1_hit = is_mask ? lct : lae(0.f + lat, lbt), q1_off = is_mask ? lct : lae(LOG_ZERO_F + lat, lbt);
    const float qt_m = is_mask ? 0.f : LOG_ZERO_F, q1_m = qt_m;
    float q[NPL];
    float qmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = j * 64 + lane;
        q[j] = lx0[j] - (c == xt ? qt_hit : qt_off);
        qmax = fmaxf(qmax, q[j]);
    }
    const float q_m = lx0_m - qt_m;
    qmax = fmaxf(wmaxf(qmax), q_m);
    float es = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) es += expf(q[j] - qmax);
    es = wsumf(es) + expf(q_m - qmax);
    const float lse = logf(es) + qmax;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = j * 64 + lane;
        const float ev = lae((q[j] - lse) + lcat1, lcbt1);
        post[j] = fminf(fmaxf(ev + (c == xt ? q1_hit : q1_off) + lse, -70.f), 0.f);
    }
    const float evm = lae((q_m - lse) + l1mcct1, lcct1);
    post_m = fminf(fmaxf(evm + q1_m + lse, -70.f), 0.f);
}

// Per grid position: kl = KL(q(x_{t-1}|x_t,x_0) || p_theta(x_{t-1}|x_t)) (:439-440), decoder_nll =
// -log p_theta(x_0|...) (:446), kl_aux = KL(x_0 || p_theta(x_0|x_t)) over the K real classes (:462).  The mask
// weights, the t == 0 switch, 1/pt and the sums over positions are a few tiny torch ops on the [B][L] outputs.
template <int NPL>
__global__ __launch_bounds__(256) void ds_loss_tail_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ x0, const int64_t* __restrict__ xt,
                                                           const int64_t* __restrict__ t, const float* __restrict__ sched,
                                                           float* __restrict__ kl, float* __restrict__ nll,
                                                           float* __restrict__ kl_aux, float* __restrict__ dbg_model,
                                                           int B, int L, int T) {
    constexpr int K = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * L) return;
    const int b = col / L, pos = col - b * L;
    float v[NPL];
    const float* lg = logits + (size_t)col * K;
#pragma unroll
    for (int j = 0; j < NPL; ++j) v[j] = lg[j * 64 + lane];
    float mx = v[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, v[j]);
    mx = wmaxf(mx);
    double se = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) se += exp((double)v[j] - (double)mx);
    const double lse64 = log(wsumd(se));
    float lp[NPL], ls[NPL];
    const int x0c = (int)x0[col], xtc = (int)xt[col], tt = (int)t[b];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        lp[j] = fminf(fmaxf((float)(((double)v[j] - (double)mx) - lse64), -70.f), 0.f);   // log p_theta(x_0 | x_t)
        ls[j] = (j * 64 + lane) == x0c ? 0.f : LOG_ZERO_F;                                 // log one-hot of x_0
    }
    const float ls_m = x0c == K ? 0.f : LOG_ZERO_F;
    float pm[NPL], pr[NPL], pm_m, pr_m;
    ds_posterior<NPL>(lp, -70.f, xtc, K, lane, sched, T + 1, tt, pm, pm_m);    // model posterior
    ds_posterior<NPL>(ls, ls_m, xtc, K, lane, sched, T + 1, tt, pr, pr_m);     // true posterior
    float a = 0.f, n = 0.f, x = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        a += expf(pr[j]) * (pr[j] - pm[j]);
        n += expf(ls[j]) * pm[j];
        x += expf(ls[j]) * (ls[j] - lp[j]);
    }
    a = wsumf(a) + expf(pr_m) * (pr_m - pm_m);
    n = wsumf(n) + expf(ls_m) * pm_m;
    x = wsumf(x);
    if (lane == 0) {
        kl[col] = a;
        nll[col] = -n;
        kl_aux[col] = x;
    }
    if (dbg_model) {
        const size_t base = (size_t)b * (K + 1) * L + pos;
#pragma unroll
        for (int j = 0; j < NPL; ++j) dbg_model[base + (size_t)(j * 64 + lane) * L] = pm[j];
        if (lane == 0) dbg_model[base + (size_t)K * L] = pm_m;
    }
}

// d(sum_b vb_loss_b) / d logits of the training loss: the first backward kernel of scope row 8f-3 (closed form and
// derivation: oracle/diffsound_oracle.py loss_tail_backward).  Same thread mapping as the forward tail.
template <int NPL>
__global__ __launch_bounds__(256) void ds_loss_tail_bwd_kernel(const float* __restrict__ logits,
                                                               const int64_t* __restrict__ x0, const int64_t* __restrict__ xt,
                                                               const int64_t* __restrict__ t, const float* __restrict__ pt,
                                                               const float* __restrict__ sched, float* __restrict__ dlogits,
                                                               int B, int L, int T, float mw_mask, float mw_other,
                                                               float aux_weight, int adaptive) {
    constexpr int K = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * L) return;
    const int b = col / L;
    float v[NPL];
    const float* lg = logits + (size_t)col * K;
#pragma unroll
    for (int j = 0; j < NPL; ++j) v[j] = lg[j * 64 + lane];
    float mx = v[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, v[j]);
    mx = wmaxf(mx);
    double se = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) se += exp((double)v[j] - (double)mx);
    const double lse64 = log(wsumd(se));
    const int x0c = (int)x0[col], xtc = (int)xt[col], tt = (int)t[b];
    const int T1 = T + 1, tm1 = (tt - 1 + T1) % T1;
    const float* S = sched;
    float lp[NPL], ls[NPL], sm[NPL];
    bool lp_live[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const float l = (float)(((double)v[j] - (double)mx) - lse64);
        sm[j] = expf(l);
        lp_live[j] = l > -70.f && l < 0.f;
        lp[j] = fminf(fmaxf(l, -70.f), 0.f);
        ls[j] = (j * 64 + lane) == x0c ? 0.f : LOG_ZERO_F;
    }
    const float ls_m = x0c == K ? 0.f : LOG_ZERO_F;
    // true posterior (forward code)
    float pr[NPL], pr_m;
    ds_posterior<NPL>(ls, ls_m, xtc, K, lane, S, T1, tt, pr, pr_m);
    // model posterior with the quantities its derivative needs
    const float lat = S[0 * T1 + tt], lbt = S[1 * T1 + tt], lct = S[2 * T1 + tt];
    const float lcat = S[4 * T1 + tt], lcbt = S[5 * T1 + tt], lcct = S[6 * T1 + tt];
    const float lcat1 = S[4 * T1 + tm1], lcbt1 = S[5 * T1 + tm1], lcct1 = S[6 * T1 + tm1], l1mcct1 = S[7 * T1 + tm1];
    const bool is_mask = xtc == K;
    const float qt_hit = is_mask ? lcct : lae(0.f + lcat, lcbt), qt_off = is_mask ? lcct : lae(LOG_ZERO_F + lcat, lcbt);
    const float q1_hit = is_mask ? lct : lae(0.f + lat, lbt), q1_off = is_mask ? lct : lae(LOG_ZERO_F + lat, lbt);
    const float qt_m = is_mask ? 0.f : LOG_ZERO_F, q1_m = qt_m;
    float q[NPL];
    float qmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        q[j] = lp[j] - ((j * 64 + lane) == xtc ? qt_hit : qt_off);
        qmax = fmaxf(qmax, q[j]);
    }
    const float q_m = -70.f - qt_m;
    qmax = fmaxf(wmaxf(qmax), q_m);
    float es = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) es += expf(q[j] - qmax);
    es = wsumf(es) + expf(q_m - qmax);
    const float lse = logf(es) + qmax;
    const float is0 = tt == 0 ? 1.f : 0.f, ipt = 1.f / pt[b];
    const float wgt = is_mask ? mw_mask : mw_other;
    const float wa = adaptive ? (float)tt / (float)T + 1.f : 1.f;
    const float nll_scale = ipt + (aux_weight != 0.f ? wa * aux_weight * ipt : 0.f);
    float w[NPL], sg[NPL], pp[NPL];
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const float A = lae((q[j] - lse) + lcat1, lcbt1);
        const float raw = A + ((j * 64 + lane) == xtc ? q1_hit : q1_off) + lse;
        sg[j] = expf((q[j] - lse) + lcat1 - A);
        pp[j] = expf(q[j] - lse);
        const bool live = raw > -70.f && raw < 0.f;
        w[j] = live ? -(1.f - is0) * expf(pr[j]) * wgt * ipt - is0 * expf(ls[j]) * nll_scale : 0.f;
        ssum += w[j] * (1.f - sg[j]);
    }
    {   // [MASK] row
        const float A = lae((q_m - lse) + l1mcct1, lcct1);
        const float raw = A + q1_m + lse;
        const float sgm = expf((q_m - lse) + l1mcct1 - A);
        const bool live = raw > -70.f && raw < 0.f;
        const float wm = live ? -(1.f - is0) * expf(pr_m) * wgt * ipt - is0 * expf(ls_m) * nll_scale : 0.f;
        ssum = wsumf(ssum) + wm * (1.f - sgm);
    }
    float glp[NPL];
    float G = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        float gq = w[j] * sg[j] + pp[j] * ssum;
        if (aux_weight != 0.f) gq -= (1.f - is0) * wa * aux_weight * ipt * wgt * expf(ls[j]);
        glp[j] = lp_live[j] ? gq : 0.f;
        G += glp[j];
    }
    G = wsumf(G);
    float* dz = dlogits + (size_t)col * K;
#pragma unroll
    for (int j = 0; j < NPL; ++j) dz[j * 64 + lane] = glp[j] - sm[j] * G;
}

extern "C" int ds_loss_tail_bwd(const float* logits, const int64_t* x0, const int64_t* xt, const int64_t* t,
                                const float* pt, const float* sched, float* dlogits, int B, int L, int K, int T,
                                float mask_weight_masked, float mask_weight_other, float aux_weight, int adaptive,
                                ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(logits && x0 && xt && t && pt && sched && dlogits && B > 0 && L > 0 && T > 0, "bad arguments");
    DS_CHECK_ARG(K == 256 || K == 512, "codebook size must be 256 or 512");
    const int cols = B * L;
    if (K == 256)
        hipLaunchKernelGGL((ds_loss_tail_bwd_kernel<4>), dim3((cols + 3) / 4), dim3(256), 0, stream, logits, x0, xt, t, pt,
                           sched, dlogits, B, L, T, mask_weight_masked, mask_weight_other, aux_weight, adaptive);
    else
        hipLaunchKernelGGL((ds_loss_tail_bwd_kernel<8>), dim3((cols + 3) / 4), dim3(256), 0, stream, logits, x0, xt, t, pt,
                           sched, dlogits, B, L, T, mask_weight_masked, mask_weight_other, aux_weight, adaptive);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_loss_tail(const float* logits, const int64_t* x0, const int64_t* xt, const int64_t* t,
                            const float* sched, float* kl, float* nll, float* kl_aux, float* dbg_model_log_prob, int B,
                            int L, int K, int T, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(logits && x0 && xt && t && sched && kl && nll && kl_aux && B > 0 && L > 0 && T > 0, "bad arguments");
    DS_CHECK_ARG(K == 256 || K == 512, "codebook size must be 256 or 512");
    const int cols = B * L;
    if (K == 256)
        hipLaunchKernelGGL((ds_loss_tail_kernel<4>), dim3((cols + 3) / 4), dim3(256), 0, stream, logits, x0, xt, t, sched,
                           kl, nll, kl_aux, dbg_model_log_prob, B, L, T);
    else
        hipLaunchKernelGGL((ds_loss_tail_kernel<8>), dim3((cols + 3) / 4), dim3(256), 0, stream, logits, x0, xt, t, sched,
                           kl, nll, kl_aux, dbg_model_log_prob, B, L, T);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- q_sample (diffusion_transformer.py:370-377): x_t ~ q(x_t | x_0) for token ids, Gumbel-argmax ----------
// log q(x_t = c | x_0) = log_add_exp(log_onehot(x_0)[c] + log_cumprod_at[t], log_cumprod_bt[t]) for the K classes,
// log_add_exp(log_onehot(x_0)[K] + log_1_min_cumprod_ct[t], log_cumprod_ct[t]) for [MASK]  (q_pred, :253-267)
__global__ __launch_bounds__(256) void ds_q_sample_kernel(const int64_t* __restrict__ x0, const int64_t* __restrict__ t,
                                                          const float* __restrict__ u, const float* __restrict__ sched,
                                                          int64_t* __restrict__ out, int B, int L, int K, int T,
                                                          const int64_t* __restrict__ gid, unsigned seed_lo,
                                                          unsigned seed_hi, int call) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * L) return;
    const int lane = threadIdx.x & 63;
    const int b = col / L, pos = col - b * L;
    const int T1 = T + 1, tt = (int)t[b];
    const float lcat = sched[4 * T1 + tt], lcbt = sched[5 * T1 + tt], lcct = sched[6 * T1 + tt],
                l1mcct = sched[7 * T1 + tt];
    const int x = (int)x0[col];
    const float hit = lae(0.f + lcat, lcbt), off = lae(LOG_ZERO_F + lcat, lcbt);
    const float* up = u ? u + (size_t)b * (K + 1) * L + pos : nullptr;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    ds_u32x4 w4 = {0u, 0u, 0u, 0u};
    for (int c = lane; c <= K; c += 64) {
        const float lq = c < K ? (c == x ? hit : off) : lae((x == K ? 0.f : LOG_ZERO_F) + l1mcct, lcct);
        float uu;
        if (up) {
            uu = up[(size_t)c * L];
        } else {                                   // stream 1 of the caption's Philox draws (ds_u01 above)
            const int j = c >> 6;
            if ((j & 3) == 0) w4 = ds_philox4x32_10((j >> 2) * 64 + (c & 63), pos | (1u << 16), call, (unsigned)gid[b], seed_lo, seed_hi);
            uu = ds_u01((j & 3) == 0 ? w4.x : (j & 3) == 1 ? w4.y : (j & 3) == 2 ? w4.z : w4.w);
        }
        const float g = -logf(-logf(uu + 1e-30f) + 1e-30f) + lq;
        if (g > best) { best = g; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[col] = bi;
}

extern "C" int ds_q_sample(const int64_t* x0, const int64_t* t, const float* u, const float* sched, int64_t* out,
                           int B, int L, int K, int T, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x0 && t && u && sched && out && B > 0 && L > 0 && K > 0 && T > 0, "bad arguments");
    hipLaunchKernelGGL(ds_q_sample_kernel, dim3((B * L + 3) / 4), dim3(256), 0, stream, x0, t, u, sched, out, B, L, K, T,
                       (const int64_t*)nullptr, 0u, 0u, 0);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_q_sample_rng(const int64_t* x0, const int64_t* t, const int64_t* gids, unsigned long long seed, int call,
                               const float* sched, int64_t* out, int B, int L, int K, int T, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x0 && t && gids && sched && out && B > 0 && L > 0 && K > 0 && T > 0, "bad arguments");
    DS_CHECK_ARG(K % 256 == 0 && L < 65536, "codebook size must be a multiple of 256, L < 65536");
    hipLaunchKernelGGL(ds_q_sample_kernel, dim3((B * L + 3) / 4), dim3(256), 0, stream, x0, t, (const float*)nullptr, sched,
                       out, B, L, K, T, gids, (unsigned)seed, (unsigned)(seed >> 32), call);
    DS_CHECK_LAUNCH();
    return 0;
}

// the uniforms the *_rng entries draw, written out in the reference's [B][K+1][L] layout (tests; same-noise comparisons
// of the two paths).  rng_stream 0: reverse steps, 1: q_sample.
__global__ __launch_bounds__(256) void ds_philox_uniforms_kernel(const int64_t* __restrict__ gid, unsigned seed_lo,
                                                                 unsigned seed_hi, int call, unsigned sid,
                                                                 float* __restrict__ u, int B, int L, int K) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= B * L) return;
    const int lane = threadIdx.x & 63;
    const int b = col / L, pos = col - b * L;
    float* up = u + (size_t)b * (K + 1) * L + pos;
    for (int g = 0; g <= K / 256; ++g) {
        const ds_u32x4 w4 = ds_philox4x32_10(g * 64 + lane, pos | (sid << 16), call, (unsigned)gid[b], seed_lo, seed_hi);
        const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (4 * g + e) * 64 + lane;
            if (c <= K) up[(size_t)c * L] = ds_u01(w[e]);
        }
    }
}
extern "C" int ds_philox_uniforms(const int64_t* gids, unsigned long long seed, int call, int rng_stream, float* u, int B,
                                  int L, int K, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(gids && u && B > 0 && L > 0 && L < 65536 && K > 0 && K % 256 == 0, "bad arguments");
    DS_CHECK_ARG(rng_stream == 0 || rng_stream == 1, "rng_stream is 0 (reverse steps) or 1 (q_sample)");
    hipLaunchKernelGGL(ds_philox_uniforms_kernel, dim3((B * L + 3) / 4), dim3(256), 0, stream, gids, (unsigned)seed,
                       (unsigned)(seed >> 32), call, (unsigned)rng_stream, u, B, L, K);
    DS_CHECK_LAUNCH();
    return 0;
}

// logits_rows >= L: rows of `logits` per sample ([B * logits_rows][K]; the denoiser's padded-row mode, api.hip)
int ds_sample_tail_rows(const float* logits, int logits_rows, const int64_t* xt, const int64_t* t, const float* u,
                        const float* sched, int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc, float* dbg_post,
                        int B, int L, int K, int T, int initial, float trunc_r, int trunc_k, ds_stream_t stream_,
                        const int64_t* gids, unsigned long long seed, int call) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(logits && xt && t && (u || gids) && sched && out_tokens, "null pointer");
    DS_CHECK_ARG(K == 256 || K == 512, "codebook size must be 256 or 512");
    DS_CHECK_ARG(trunc_k >= 0 && !(trunc_k > 0 && trunc_r >= 0.f), "top-k and top-r truncation are exclusive");
    DS_CHECK_ARG(logits_rows >= L && L < 65536, "logits rows per sample");
    SampleParams p{logits, xt, t, u, sched, out_tokens, dbg_log_pred, dbg_trunc, dbg_post, B, L, T, initial, trunc_r,
                   trunc_k, logits_rows};
    const SampleRng g{gids, (unsigned)seed, (unsigned)(seed >> 32), call};
    const int cols = B * L;
    const dim3 grid((cols + 3) / 4);
    if (K == 256 && u) hipLaunchKernelGGL((ds_sample_tail_kernel<4, false>), grid, dim3(256), 0, stream, p, g);
    else if (K == 256) hipLaunchKernelGGL((ds_sample_tail_kernel<4, true>), grid, dim3(256), 0, stream, p, g);
    else if (u) hipLaunchKernelGGL((ds_sample_tail_kernel<8, false>), grid, dim3(256), 0, stream, p, g);
    else hipLaunchKernelGGL((ds_sample_tail_kernel<8, true>), grid, dim3(256), 0, stream, p, g);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_sample_tail_ex(const float* logits, const int64_t* xt, const int64_t* t, const float* u,
                                 const float* sched, int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc,
                                 float* dbg_post, int B, int L, int K, int T, int initial, float trunc_r,
                                 int trunc_k, ds_stream_t stream) {
    DS_CHECK_ARG(u, "null pointer");
    return ds_sample_tail_rows(logits, L, xt, t, u, sched, out_tokens, dbg_log_pred, dbg_trunc, dbg_post, B, L, K, T, initial,
                               trunc_r, trunc_k, stream, nullptr, 0ull, 0);
}

extern "C" int ds_sample_tail_rng(const float* logits, const int64_t* xt, const int64_t* t, const int64_t* gids,
                                  unsigned long long seed, int call, const float* sched, int64_t* out_tokens, int B,
                                  int L, int K, int T, int initial, float trunc_r, int trunc_k, ds_stream_t stream) {
    DS_CHECK_ARG(gids, "null pointer");
    return ds_sample_tail_rows(logits, L, xt, t, nullptr, sched, out_tokens, nullptr, nullptr, nullptr, B, L, K, T, initial,
                               trunc_r, trunc_k, stream, gids, seed, call);
}

extern "C" int ds_sample_tail(const float* logits, const int64_t* xt, const int64_t* t, const float* u,
                              const float* sched, int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc,
                              float* dbg_post, int B, int L, int K, int T, int initial, float trunc_r,
                              ds_stream_t stream) {
    return ds_sample_tail_ex(logits, xt, t, u, sched, out_tokens, dbg_log_pred, dbg_trunc, dbg_post, B, L, K, T, initial,
                             trunc_r, 0, stream);
}
