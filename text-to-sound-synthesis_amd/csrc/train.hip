// Row / elementwise kernels of the training step (scope row 8f-3): the backward of AdaLN / LayerNorm, GELU2,
// row softmax, the embedding, column sums for bias / scale gradients, and the fused AdamW update.  All HBM-bound,
// one wave per 1024-wide row where rows are reduced.  The GEMM-shaped parts of the backward run on the GEMM kernels.
// Reference ops: AdaLayerNorm / nn.LayerNorm / GELU2 (transformer_utils.py:111-149), DalleMaskImageEmbedding
// (dalle_mask_image_embedding.py:36-58), torch.optim.AdamW as configured by configs/caps.yaml:111-115.
#include "common.h"

__device__ __forceinline__ float tr_wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- LayerNorm / AdaLN backward ---------------------------------------------------------------------------
// forward  xn = (x - mean) * rstd;  y = xn * s + b  with  s = 1 + tab[t][c] (AdaLN, mode 0) or gamma[c] (mode 1)
// backward g = dy * s;  dx = rstd * (g - mean(g) - xn * mean(g * xn));  dyxn = dy * xn  (for d scale, via ds_colsum)
template <int D>
__global__ __launch_bounds__(256) void ds_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, float* __restrict__ dyxn, int M,
                                                               int L, int mode, const float* __restrict__ tab,
                                                               const int64_t* __restrict__ t,
                                                               const float* __restrict__ gamma, float eps, int accumulate) {
    constexpr int NV = D / 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    f32x4 v[NV], g[NV];
    const float* xr = x + (size_t)row * D;
    const float* dr = dy + (size_t)row * D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j] = *(const f32x4*)(xr + (j * 64 + lane) * 4);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = tr_wsum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = v[j][k] - mean;
            q += d * d;
        }
    const float rstd = 1.f / sqrtf(tr_wsum(q) * (1.f / D) + eps);
    const float* sc = mode == 0 ? tab + (size_t)t[row / L] * 2 * D : gamma;
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        const f32x4 a = *(const f32x4*)(sc + c), d4 = *(const f32x4*)(dr + c);
        f32x4 px;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xn = (v[j][k] - mean) * rstd;
            v[j][k] = xn;
            g[j][k] = d4[k] * (mode == 0 ? 1.f + a[k] : a[k]);
            px[k] = d4[k] * xn;
            sg += g[j][k];
            sgx += g[j][k] * xn;
        }
        if (dyxn) *(f32x4*)(dyxn + (size_t)row * D + c) = px;
    }
    const float mg = tr_wsum(sg) * (1.f / D), mgx = tr_wsum(sgx) * (1.f / D);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rstd * (g[j][k] - mg - v[j][k] * mgx);
        if (accumulate) o += *(const f32x4*)(dx + (size_t)row * D + c);     // the residual stream's gradient: dx += d norm-input
        *(f32x4*)(dx + (size_t)row * D + c) = o;
    }
}

static int ln_bwd_launch(const float* x, const float* dy, float* dx, float* dyxn, int M, int L, int D, int mode, const float* table,
                         const int64_t* t, const float* gamma, int accumulate, ds_stream_t stream) {
    DS_CHECK_ARG(x && dy && dx && M > 0 && D == 1024, "bad arguments (D = 1024 is built)");
    DS_CHECK_ARG(mode == 0 ? (table && t && L > 0) : (mode == 1 && gamma), "mode 0 needs table / t / L, mode 1 gamma");
    hipLaunchKernelGGL((ds_layernorm_bwd_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, dyxn,
                       M, mode == 0 ? L : 1, mode, table, t, gamma, 1e-5f, accumulate);
    DS_CHECK_LAUNCH();
    return 0;
}
extern "C" int ds_layernorm_bwd(const float* x, const float* dy, float* dx, float* dyxn, int M, int L, int D, int mode,
                                const float* table, const int64_t* t, const float* gamma, ds_stream_t stream) {
    return ln_bwd_launch(x, dy, dx, dyxn, M, L, D, mode, table, t, gamma, 0, stream);
}
// ---- the same backward with the scale / shift gradient sums folded in (round 5) ------------------------------------------
// d scale = column sums of dy * xn, d shift = column sums of dy, per sample (AdaLN) or over all rows (LayerNorm).  The form
// above writes dy * xn to HBM for ds_colsum_ws (one write + two reads of an [M][D] matrix and four launches per norm); here a
// workgroup walks a CHUNK of <= LNB_CHUNK consecutive rows of one group, its four waves keep the two sums of their rows'
// columns in registers, combine them through LDS and write ONE partial row pair part[chunk][2][D]; ds_colsum adds the chunks of
// a group in a fixed order (deterministic).  chunks per group = ceil(rows per group / LNB_CHUNK), rows per chunk = ceil(rows /
// chunks).  LNB_CHUNK = 16: at B = 20 that is 340 workgroups (64-row chunks were 100 workgroups on 256 CUs: 61 us per launch
// against 18 us for the plain kernel, profiles/r05j_train_kernel_top.txt).
#define LNB_CHUNK 16
extern "C" int ds_layernorm_bwd_chunks(int M, int L, int mode) {
    const int rows = mode == 0 ? L : M;
    return rows > 0 ? (rows + LNB_CHUNK - 1) / LNB_CHUNK : 0;
}
template <int D>
__global__ __launch_bounds__(256) void ds_layernorm_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, float* __restrict__ part, int M, int L,
                                                                    int mode, const float* __restrict__ tab,
                                                                    const int64_t* __restrict__ t, const float* __restrict__ gamma,
                                                                    float eps, int accumulate, int rows_per_group, int cpg, int rpc) {
    constexpr int NV = D / 256;
    __shared__ float red[4][2][D];
    const int grp = blockIdx.x / cpg, ck = blockIdx.x - grp * cpg;
    const int g0 = grp * rows_per_group;
    const int r_lo = g0 + ck * rpc;
    int r_hi = r_lo + rpc;
    if (r_hi > g0 + rows_per_group) r_hi = g0 + rows_per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* sc = mode == 0 ? tab + (size_t)t[grp] * 2 * D : gamma;      // (mode 0: one group = one sample)
    f32x4 a[NV], sx[NV], sd[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        a[j] = *(const f32x4*)(sc + (j * 64 + lane) * 4);
        sx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        sd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = r_lo + wave; row < r_hi; row += 4) {
        f32x4 v[NV], g[NV];
        const float* xr = x + (size_t)row * D;
        const float* dr = dy + (size_t)row * D;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *(const f32x4*)(xr + (j * 64 + lane) * 4);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = tr_wsum(s) * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[j][k] - mean;
                q += d * d;
            }
        const float rstd = 1.f / sqrtf(tr_wsum(q) * (1.f / D) + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 d4 = *(const f32x4*)(dr + (j * 64 + lane) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xn = (v[j][k] - mean) * rstd;
                v[j][k] = xn;
                g[j][k] = d4[k] * (mode == 0 ? 1.f + a[j][k] : a[j][k]);
                sx[j][k] += d4[k] * xn;
                sd[j][k] += d4[k];
                sg += g[j][k];
                sgx += g[j][k] * xn;
            }
        }
        const float mg = tr_wsum(sg) * (1.f / D), mgx = tr_wsum(sgx) * (1.f / D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (j * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rstd * (g[j][k] - mg - v[j][k] * mgx);
            if (accumulate) o += *(const f32x4*)(dx + (size_t)row * D + c);
            *(f32x4*)(dx + (size_t)row * D + c) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        *(f32x4*)(&red[wave][0][(j * 64 + lane) * 4]) = sx[j];
        *(f32x4*)(&red[wave][1][(j * 64 + lane) * 4]) = sd[j];
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, c = i - which * D;
        out[i] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}
// part: [G * chunks][2][D] floats with G = M / L (mode 0) or 1 (mode 1), chunks = ds_layernorm_bwd_chunks(M, L, mode);
// accumulate != 0: dx += (the residual connection), else dx =.
extern "C" int ds_layernorm_bwd_sums(const float* x, const float* dy, float* dx, float* part, int M, int L, int D, int mode,
                                     const float* table, const int64_t* t, const float* gamma, int accumulate, ds_stream_t stream) {
    DS_CHECK_ARG(x && dy && dx && part && M > 0 && D == 1024, "bad arguments (D = 1024 is built)");
    DS_CHECK_ARG(mode == 0 ? (table && t && L > 0 && M % L == 0) : (mode == 1 && gamma), "mode 0 needs table / t / L | M, mode 1 gamma");
    const int rows = mode == 0 ? L : M, G = mode == 0 ? M / L : 1;
    const int cpg = (rows + LNB_CHUNK - 1) / LNB_CHUNK, rpc = (rows + cpg - 1) / cpg;
    hipLaunchKernelGGL((ds_layernorm_bwd_sums_kernel<1024>), dim3(G * cpg), dim3(256), 0, (hipStream_t)stream, x, dy, dx, part, M,
                       mode == 0 ? L : 1, mode, table, t, gamma, 1e-5f, accumulate, rows, cpg, rpc);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- column sums:  out[g][c] (+)= sum_{r < R} x[(g * R + r) * ld + c]   (bias / scale / embedding gradients) ----------
// grid (ceil(C/256), G); each thread owns one column and walks the rows: coalesced, deterministic order
__global__ __launch_bounds__(256) void ds_colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                                        long long ld, long long gstride, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = x + (size_t)blockIdx.y * gstride + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 0;
    for (; r + 3 < R; r += 4) {
        s0 += p[(size_t)r * ld];
        s1 += p[(size_t)(r + 1) * ld];
        s2 += p[(size_t)(r + 2) * ld];
        s3 += p[(size_t)(r + 3) * ld];
    }
    for (; r < R; ++r) s0 += p[(size_t)r * ld];
    const float s = (s0 + s1) + (s2 + s3);
    float* o = out + (size_t)blockIdx.y * C + c;
    *o = accumulate ? *o + s : s;
}

// The same sums, same order, for launches too small to hide a chain of R dependent row visits (the bias gradients: 83 row-tile
// partials x 1024 .. 4096 columns were 4 .. 16 workgroups walking 83 rows each, ~8 us a call, 133 calls per training
// iteration): the four interleaved chains s0 .. s3 of a column go to four threads (one per wave: a wave still reads 64
// consecutive floats), combined through LDS as (s0 + s1) + (s2 + s3); the R % 4 tail rows join chain 0 as above.
__global__ __launch_bounds__(256) void ds_colsum4_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                                         long long ld, long long gstride, int accumulate) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < C) {
        const float* p = x + (size_t)blockIdx.y * gstride + c;
        const int R4 = R & ~3;
        int r = ch;
        // a chain's loads go out in batches (16, then 4, then single rows) and are added in row order: with one load per loop
        // trip every row cost a full memory latency (~0.45 us: 83 rows = 10 us a call)
#define CS4_BATCH(NB)                                                      \
        for (; r + 4 * (NB - 1) < R4; r += 4 * NB) {                       \
            float v[NB];                                                   \
            _Pragma("unroll") for (int k = 0; k < NB; ++k) v[k] = p[(size_t)(r + 4 * k) * ld]; \
            _Pragma("unroll") for (int k = 0; k < NB; ++k) s += v[k];      \
        }
        CS4_BATCH(16)
        CS4_BATCH(4)
        CS4_BATCH(1)
#undef CS4_BATCH
        if (ch == 0)
            for (int r2 = R4; r2 < R; ++r2) s += p[(size_t)r2 * ld];
    }
    part[ch][cx] = s;
    __syncthreads();
    if (ch == 0 && c < C) {
        const float t = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
        float* o = out + (size_t)blockIdx.y * C + c;
        *o = accumulate ? *o + t : t;
    }
}

extern "C" int ds_colsum(const float* x, float* out, int G, int R, int C, long long ld, long long gstride, int accumulate,
                         ds_stream_t stream) {
    DS_CHECK_ARG(x && out && G > 0 && R > 0 && C > 0 && ld >= C, "bad arguments");
    if (R >= 16 && (long long)((C + 255) / 256) * G < 256) {      // few workgroups, long chains: one thread per chain
        hipLaunchKernelGGL(ds_colsum4_kernel, dim3((C + 63) / 64, G), dim3(256), 0, (hipStream_t)stream, x, out, R, C, ld, gstride,
                           accumulate);
        DS_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(ds_colsum_kernel, dim3((C + 255) / 256, G), dim3(256), 0, (hipStream_t)stream, x, out, R, C, ld, gstride,
                       accumulate);
    DS_CHECK_LAUNCH();
    return 0;
}

// The same sums for TALL inputs (bias / LayerNorm gradients: R = B * L rows of 1024 .. 4096 columns, where one thread per
// column is 4 .. 16 workgroups walking thousands of rows: measured 600 us per call, 42 % of a training step): the rows are
// cut into `rs` chunks summed by separate workgroups into work[g][chunk][c] (stage 1, grid z = chunk), and the kernel
// above adds the chunks in a fixed order (stage 2) -- deterministic, no atomics.  work: G * rs * C floats, rs <= 64.
__global__ __launch_bounds__(256) void ds_colsum_chunk_kernel(const float* __restrict__ x, float* __restrict__ work, int R, int C,
                                                              long long ld, long long gstride, int chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.z * chunk;
    int r1 = r0 + chunk;
    if (r1 > R) r1 = R;
    const float* p = x + (size_t)blockIdx.y * gstride + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += p[(size_t)r * ld];
        s1 += p[(size_t)(r + 1) * ld];
        s2 += p[(size_t)(r + 2) * ld];
        s3 += p[(size_t)(r + 3) * ld];
    }
    for (; r < r1; ++r) s0 += p[(size_t)r * ld];
    work[((size_t)blockIdx.y * gridDim.z + blockIdx.z) * C + c] = (s0 + s1) + (s2 + s3);
}

// ---- few rows times a row-major matrix:  part[ks][g][b][d] = sum_{k in range ks} x[g][b][k] * W[g][k][d]  (b < B <= 32) -------
// The AdaLN backward's  d silu-input = (d modulation) W  for all 2 n_layer modules (W = linear.weight [2D][D],
// transformer_utils.py:145-147): B = 20 rows against 38 matrices of 8 MB.  As a GEMM it needs W^T as its K-contiguous operand -- a
// 319 MB transposing copy per iteration plus a tile program that is 84 % padding rows.  Here a thread owns one output column
// d and walks its k-range: W is read once, coalesced along d, straight from the parameters; the B rows of x sit in LDS and
// are read as broadcasts (16 bytes = 4 k per read).  The KS k-ranges are separate workgroups (parallelism: 4 KS G of them);
// their partial results are added by ds_colsum in a fixed order.  Bound by the 2 D D 4 bytes of each matrix.
#define RM_KC 256                       // k per workgroup
__global__ __launch_bounds__(256) void ds_rows_times_matrix_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                   float* __restrict__ part, int B, int K, int D) {
    __shared__ __attribute__((aligned(16))) float xs[32][RM_KC];
    const int g = blockIdx.z, ks = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    const int k0 = ks * RM_KC;
    const float* xg = x + ((size_t)g * B) * K + k0;
    for (int i = threadIdx.x; i < B * (RM_KC / 4); i += 256) {
        const int b = i / (RM_KC / 4), c4 = (i - b * (RM_KC / 4)) * 4;
        *(f32x4*)&xs[b][c4] = *(const f32x4*)(xg + (size_t)b * K + c4);
    }
    __syncthreads();
    if (d >= D) return;
    float acc[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) acc[b] = 0.f;
    const float* wp = W + ((size_t)g * K + k0) * D + d;
    for (int k = 0; k < RM_KC; k += 8) {                     // eight rows of W in flight per thread (a load per trip would
        float w[8];                                          //  cost a memory latency per row: see ds_colsum4_kernel)
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = wp[(size_t)(k + u) * D];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            if (b < B) {                                     // (uniform; B is small and fixed per launch)
                const f32x4 xa = *(const f32x4*)&xs[b][k], xb = *(const f32x4*)&xs[b][k + 4];
                float a = acc[b];
#pragma unroll
                for (int u = 0; u < 4; ++u) a = fmaf(xa[u], w[u], a);
#pragma unroll
                for (int u = 0; u < 4; ++u) a = fmaf(xb[u], w[4 + u], a);
                acc[b] = a;
            }
        }
    }
    float* o = part + (((size_t)ks * gridDim.z + g) * B) * D + d;
#pragma unroll
    for (int b = 0; b < 32; ++b)
        if (b < B) o[(size_t)b * D] = acc[b];
}

// x [G][B][K], W [G][K][D] (row-major), part [K / 256][G][B][D]: the caller adds the K / 256 partial results (ds_colsum).
extern "C" int ds_rows_times_matrix(const float* x, const float* W, float* part, int G, int B, int K, int D, ds_stream_t stream) {
    DS_CHECK_ARG(x && W && part && G > 0 && B > 0 && B <= 32 && K > 0 && K % RM_KC == 0 && D > 0, "B <= 32 rows, K a multiple of 256");
    DS_CHECK_ARG((((uintptr_t)x) & 15) == 0 && K % 4 == 0, "x must be 16-byte aligned");
    DS_CHECK_ARG(G <= 65535 && K / RM_KC <= 65535, "grid limits");
    hipLaunchKernelGGL(ds_rows_times_matrix_kernel, dim3((D + 255) / 256, K / RM_KC, G), dim3(256), 0, (hipStream_t)stream, x, W, part, B,
                       K, D);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- sum of B outer products:  out[g][n][d] = sum_{b < B} a[g][b][n] * s[g][b][d]   (B <= 32) --------------------------------
// The AdaLN backward's weight gradient  d linear.weight = (d modulation)^T silu(emb(t_b))  for all modules: [2D][D] outputs from a
// contraction over the B = 20 samples.  As a GEMM (K = 32 after padding) it was one k-tile of prologue + epilogue per 128 x 128
// tile, 308 us for 319 MB of output; here a thread keeps its four columns of the B rows of s in registers and walks 32
// output rows, whose B coefficients sit in LDS: one pass of 16-byte stores, bound by the output bytes.
__global__ __launch_bounds__(256) void ds_rows_outer_kernel(const float* __restrict__ a, const float* __restrict__ s,
                                                            float* __restrict__ out, int B, int N, int D) {
    __shared__ float as[32][33];                      // [n - n0][b]
    const int g = blockIdx.y, n0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * B; i += 256) {
        const int b = i >> 5, nn = i & 31;            // (consecutive threads read consecutive n of one sample row)
        as[nn][b] = n0 + nn < N ? a[((size_t)g * B + b) * N + n0 + nn] : 0.f;
    }
    __syncthreads();
    for (int d4 = threadIdx.x * 4; d4 < D; d4 += 1024) {
        f32x4 sv[32];
#pragma unroll
        for (int b = 0; b < 32; ++b)
            if (b < B) sv[b] = *(const f32x4*)(s + ((size_t)g * B + b) * D + d4);
        for (int nn = 0; nn < 32 && n0 + nn < N; ++nn) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 32; ++b)
                if (b < B) acc += as[nn][b] * sv[b];
            *(f32x4*)(out + ((size_t)g * N + n0 + nn) * D + d4) = acc;
        }
    }
}

// a [G][B][N], s [G][B][D] -> out [G][N][D];  B <= 32, D % 4 == 0, 16-byte aligned s / out
extern "C" int ds_rows_outer(const float* a, const float* s, float* out, int G, int B, int N, int D, ds_stream_t stream) {
    DS_CHECK_ARG(a && s && out && G > 0 && G <= 65535 && B > 0 && B <= 32 && N > 0 && D > 0 && D % 4 == 0, "B <= 32 rows, D % 4 == 0");
    DS_CHECK_ARG(((((uintptr_t)s) | ((uintptr_t)out)) & 15) == 0, "s / out must be 16-byte aligned");
    hipLaunchKernelGGL(ds_rows_outer_kernel, dim3((N + 31) / 32, G), dim3(256), 0, (hipStream_t)stream, a, s, out, B, N, D);
    DS_CHECK_LAUNCH();
    return 0;
}

// number of row chunks ds_colsum_ws uses for (G, R, C): about 2048 workgroups in stage 1, >= 16 rows per chunk, <= 64
static int ds_colsum_chunks(int G, int R, int C) {
    const long blocks1 = (long)((C + 255) / 256) * G;
    long rs = 2048 / (blocks1 > 0 ? blocks1 : 1);
    if (rs > 64) rs = 64;
    if (rs > (R + 15) / 16) rs = (R + 15) / 16;
    return rs < 1 ? 1 : (int)rs;
}

extern "C" int ds_colsum_ws(const float* x, float* out, int G, int R, int C, long long ld, long long gstride, int accumulate,
                            float* work, long long work_floats, ds_stream_t stream) {
    DS_CHECK_ARG(x && out && G > 0 && R > 0 && C > 0 && ld >= C, "bad arguments");
    const int rs = ds_colsum_chunks(G, R, C);
    if (rs == 1 || !work) return ds_colsum(x, out, G, R, C, ld, gstride, accumulate, stream);
    DS_CHECK_ARG(work_floats >= (long long)G * rs * C, "work: G * 64 * C floats are always enough");
    const int chunk = (R + rs - 1) / rs;
    hipLaunchKernelGGL(ds_colsum_chunk_kernel, dim3((C + 255) / 256, G, rs), dim3(256), 0, (hipStream_t)stream, x, work, R, C, ld,
                       gstride, chunk);
    DS_CHECK_LAUNCH();
    return ds_colsum(work, out, G, rs, C, C, (long long)rs * C, accumulate, stream);
}

// ---- GELU2 (x * sigmoid(1.702 x), transformer_utils.py:111-115) forward and backward, elementwise ---------------------
__global__ __launch_bounds__(256) void ds_gelu2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ out, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 v = *(const f32x4*)(x + i);
    f32x4 o;
    if (dy) {
        const f32x4 d = *(const f32x4*)(dy + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = 1.f / (1.f + expf(-1.702f * v[k]));
            o[k] = d[k] * (sg + 1.702f * v[k] * sg * (1.f - sg));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] / (1.f + expf(-1.702f * v[k]));
    }
    *(f32x4*)(out + i) = o;
}

// dy == NULL: out = gelu2(x);  else out = dy * gelu2'(x).  n % 4 == 0.
extern "C" int ds_gelu2(const float* x, const float* dy, float* out, long long n, ds_stream_t stream) {
    DS_CHECK_ARG(x && out && n > 0 && n % 4 == 0, "bad arguments (n % 4 == 0)");
    hipLaunchKernelGGL(ds_gelu2_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- row softmax backward, in place:  dS = scale * P * (dP - sum_j dP_j P_j)  over the first n columns of a row -------
__global__ __launch_bounds__(256) void ds_softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                                  int rows, int n, int ld, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* pr = P + (size_t)row * ld;
    float* dr = dP + (size_t)row * ld;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += dr[c] * pr[c];
    s = tr_wsum(s);
    for (int c = lane; c < ld; c += 64) dr[c] = c < n ? scale * pr[c] * (dr[c] - s) : 0.f;
}

extern "C" int ds_softmax_bwd_rows(const float* P, float* dP, int rows, int n, int ld, float scale, ds_stream_t stream) {
    DS_CHECK_ARG(P && dP && rows > 0 && n > 0 && ld >= n, "bad arguments");
    hipLaunchKernelGGL(ds_softmax_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, P, dP, rows, n, ld,
                       scale);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- embedding backward: d emb[token[m]][:] += dx[m][:]  (fp32 atomics; the content embedding has 257 rows) ------------
__global__ __launch_bounds__(256) void ds_embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ tok,
                                                           float* __restrict__ demb, int M, int D, int rows) {
    const int m = blockIdx.x;
    long long tk = tok[m];
    if (tk < 0 || tk >= rows) return;
    for (int c = threadIdx.x; c < D; c += 256) atomicAdd(demb + (size_t)tk * D + c, dx[(size_t)m * D + c]);
}

extern "C" int ds_embed_bwd(const float* dx, const int64_t* tokens, float* demb, int M, int D, int rows, ds_stream_t stream) {
    DS_CHECK_ARG(dx && tokens && demb && M > 0 && D > 0 && rows > 0, "bad arguments");
    hipLaunchKernelGGL(ds_embed_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, dx, tokens, demb, M, D, rows);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments), one fused pass ----------------
__global__ __launch_bounds__(256) void ds_adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                       float b1, float b2, float eps, float wd, float bc1, float bc2s) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float pi = p[i] * (1.f - lr * wd);
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi;
}

// step >= 1: the number of this update (bias corrections 1 - beta^step)
extern "C" int ds_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, ds_stream_t stream) {
    DS_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(ds_adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2s);
    DS_CHECK_LAUNCH();
    return 0;
}



// ---- max |x| into *out (caller zeroes it): non-negative floats order like their bit patterns -------------------------
// ONE atomic per workgroup, at most 512 workgroups, 16-byte loads.  (Round 3: one atomic per WAVE of 2048 workgroups -- 8192
// serialised atomics on one address, 99 us per call, 134 calls per training iteration = 13 % of it.)
__global__ __launch_bounds__(256) void ds_amax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    __shared__ float wm[4];
    float m = 0.f;
    const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
    if ((((uintptr_t)x) & 15) == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f32x4 v = *(const f32x4*)(x + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fabsf(v[e]);
                m = a > m ? a : m;            // NaN never wins: a calibration quantity, not a validity check
            }
        }
        for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float a = fabsf(x[i]);
            m = a > m ? a : m;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float a = fabsf(x[i]);
            m = a > m ? a : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wm[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = wm[w] > b ? wm[w] : b;
        if (b > 0.f) atomicMax(out, __float_as_uint(b));
    }
}

extern "C" int ds_amax(const float* x, long long n, float* out, ds_stream_t stream) {
    DS_CHECK_ARG(x && out && n > 0, "bad arguments");
    const long long blocks = (n + 4095) / 4096;          // >= 16 elements per thread
    hipLaunchKernelGGL(ds_amax_kernel, dim3((unsigned)(blocks < 512 ? (blocks > 0 ? blocks : 1) : 512)), dim3(256), 0, (hipStream_t)stream, x, n,
                       (unsigned*)out);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- AdamW with the step's scalars in device memory: hyper = { lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale } ----
// (a captured hipGraph replays the same kernel arguments every iteration; the learning rate, the bias corrections and
// the clip coefficient of the iteration are therefore read from a 4-float device buffer the host refreshes)
__global__ __launch_bounds__(256) void ds_adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long long n,
                                                           const float* __restrict__ hyper, float b1, float b2, float eps,
                                                           float wd) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2];
    const float gi = g[i] * hyper[3];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float pi = p[i] * (1.f - lr * wd);
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi;
}

extern "C" int ds_adamw_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                            float beta2, float eps, float weight_decay, ds_stream_t stream) {
    DS_CHECK_ARG(p && g && m && v && hyper && n > 0, "bad arguments");
    hipLaunchKernelGGL(ds_adamw_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       hyper, beta1, beta2, eps, weight_decay);
    DS_CHECK_LAUNCH();
    return 0;
}
