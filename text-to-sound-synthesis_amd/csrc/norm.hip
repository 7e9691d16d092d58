// Row-wise kernels of the denoiser (HBM-bound; one wavefront per 1024-wide row, float4 loads):
//   ds_embed       token + position embedding            (dalle_mask_image_embedding.py:36-58)
//   ds_layernorm   LayerNorm(eps 1e-5) with either the AdaLN modulation  y = xn*(1+scale[t])+shift[t]
//                  (transformer_utils.py:134-149, scale/shift tabulated per timestep at load time)
//                  or the ordinary affine  y = xn*gamma + beta  (ln2 / to_logits.0).
// and GroupNorm(32, C, eps 1e-6) statistics for the SpecVQGAN decoder
// (specvqgan/modules/diffusionmodules/model.py:34-35), emitted as the per-(sample, channel)
// affine  a' = a*scale + shift  that the conv loaders apply on the fly (gemm_f32.hip).
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// x[m][:] = emb[tok[m]][:] + pos[m % L][:]          D = 1024
// Lp >= L: output rows per sample (the denoiser's padded-row mode: sample b occupies rows b Lp .. b Lp + Lp - 1, the rows
// past position L - 1 are written as zeros); tokens stay [B][L]
__global__ __launch_bounds__(256) void ds_embed_kernel(const int64_t* __restrict__ tok,
                                                       const float* __restrict__ emb,
                                                       const float* __restrict__ pos,
                                                       float* __restrict__ out, int M, int L, int D, int f16, int Lp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const int b = row / Lp, ps = row - b * Lp;
    float* o = out + (size_t)row * D;
    if (ps >= L) {
        for (int c = lane * 4; c < D; c += 256) *(f32x4*)(o + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    long long t = tok[(size_t)b * L + ps];
    if (t < 0) t = 0;  // index[index < 0] = 0, dalle_mask_image_embedding.py:41
    const float* e = emb + (size_t)t * D;
    const float* p = pos + (size_t)ps * D;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(e + c), b = *(const f32x4*)(p + c);
        f32x4 r = a + b;
        if (f16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = ds_r16(ds_r16(a[k]) + ds_r16(b[k]));
        }
        *(f32x4*)(o + c) = r;
    }
}

// mode 0: AdaLN   y = xn * (1 + tab[t[b]][c]) + tab[t[b]][D + c],  b = row / L
// mode 1: affine  y = xn * gamma[c] + beta[c]
template <int D>
__global__ __launch_bounds__(256) void ds_layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int M, int L, int mode,
                                                           const float* __restrict__ tab,  // [T][2D]
                                                           const int64_t* __restrict__ t,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int f16) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    constexpr int NV = D / 256;
    f32x4 v[NV];
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j] = *(const f32x4*)(xr + (j * 64 + lane) * 4);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = v[j][k] - mean;
            q += d * d;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / D) + eps);  // biased variance, as torch
    const float *sc, *sh;
    if (mode == 0) {
        const long long tt = t[row / L];
        sc = tab + (size_t)tt * 2 * D;
        sh = sc + D;
    } else {
        sc = gamma;
        sh = beta;
    }
    float* yr = y + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        const f32x4 a = *(const f32x4*)(sc + c), b = *(const f32x4*)(sh + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xn = (v[j][k] - mean) * rstd;
            o[k] = mode == 0 ? xn * (1.f + a[k]) + b[k] : xn * a[k] + b[k];
            if (f16 == 1) o[k] = ds_r16(o[k]);
        }
        if (f16 == 2) {  // packed split planes (common.h ds_packed_off) for the f16x2 GEMM's A operand
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = ds_split_hi(o[k]);
                lo[k] = ds_split_lo(o[k], hi[k]);
            }
            _Float16* yh = (_Float16*)y + ds_packed_off(row, c, D / 32);   // 4 halves inside one 16-byte chunk
            *(h4*)yh = hi;
            *(h4*)(yh + (size_t)((M + 15) & ~15) * D) = lo;
        } else {
            *(f32x4*)(yr + c) = o;
        }
    }
}

// ---- GroupNorm statistics over a channels-last tensor x[B][P][C] (P = H*W pixels) -----------
// pass 1: grid (chunks, B): each block reduces PCHUNK pixels into per-channel double partials
// pass 2: grid (B): combine chunks -> per-group mean/rstd -> scale/shift per channel.
#define GN_PCHUNK 256
// A block reduces GN_PCHUNK pixels x C channels: thread = (pixel lane, 4 consecutive channels as one float4); the pixel lanes
// of a channel quad are added up through LDS.  (The first version gave a thread ONE channel and walked the 256 pixels in a
// dependent load -> double-add chain: 126-260 us per call on tensors that take 1-2 us to read.)  C % 4 == 0, C <= 1024.
__global__ __launch_bounds__(256) void ds_gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                            int P, int C) {
    // part[b][chunk][2][C]
    __shared__ double red[256][8];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int p0 = chunk * GN_PCHUNK;
    const int p1 = min(P, p0 + GN_PCHUNK);
    const float* xb = x + ((size_t)b * P) * C;
    const int quads = C >> 2;                       // float4 columns
    const int lanes = 256 / quads > 0 ? 256 / quads : 1;   // pixel lanes per pass over the channel quads
    double* o = part + (((size_t)b * nchunk + chunk) * 2) * C;
    for (int q0 = 0; q0 < quads; q0 += 256) {       // (C > 1024 would take more than one pass; not used)
        const int qd = q0 + (int)threadIdx.x % (quads < 256 ? quads : 256);
        const int pl = (int)threadIdx.x / (quads < 256 ? quads : 256);
        double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
        if (qd < quads && pl < lanes) {
            for (int p = p0 + pl; p < p1; p += lanes) {
                const f32x4 v = *(const f32x4*)(xb + (size_t)p * C + qd * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = (double)v[e];
                    s[e] += d;
                    q[e] += d * d;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = q[e]; }
        __syncthreads();
        if (pl == 0 && qd < quads) {
            for (int l = 1; l < lanes; ++l)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + l * quads][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[qd * 4 + e] = red[threadIdx.x][e];
                o[C + qd * 4 + e] = red[threadIdx.x][4 + e];
            }
        }
        __syncthreads();
    }
}

// grid (groups, B): one block per (sample, group) adds that group's partial sums over the chunks (fixed order: thread t
// takes items t, t + 256, ..., then a tree over the block) and writes the group's channels.  (Rounds 1-3 ran ONE block per
// sample over all groups -- 48 us per call once ds_conv3x3_f16x2 hands over 540 tile partials per sample.)
__global__ __launch_bounds__(256) void ds_gn_finish_kernel(const double* __restrict__ part, int nchunk, int P, int C,
                                                           int groups, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nchunk * cpg; i += 256) {
        const int ch = i / cpg, c = g * cpg + (i - ch * cpg);
        const double* o = part + (((size_t)b * nchunk + ch) * 2) * C;
        s += o[c];
        q += o[C + c];
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            rs[threadIdx.x] += rs[threadIdx.x + w];
            rq[threadIdx.x] += rq[threadIdx.x + w];
        }
        __syncthreads();
    }
    const double n = (double)P * cpg;
    const double mean = rs[0] / n;
    double var = rq[0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += 256) {
        const double ga = gamma[c], be = beta[c];
        scale[(size_t)b * C + c] = (float)(rstd * ga);
        shift[(size_t)b * C + c] = (float)(be - mean * rstd * ga);
    }
}

// ---- C ABI entry points ---------------------------------------------------------------------
extern "C" int ds_embed(const int64_t* tokens, const float* emb, const float* pos, float* out, int M, int L,
                        int D, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(tokens && emb && pos && out, "null pointer");
    DS_CHECK_ARG(M > 0 && L > 0 && D % 4 == 0, "bad shape");
    hipLaunchKernelGGL(ds_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tokens, emb, pos, out, M, L, D, 0, L);
    DS_CHECK_LAUNCH();
    return 0;
}

// the same with Lp >= L output rows per sample (rows past position L - 1 are zeros): x [B * Lp][D]
int ds_embed_rows(const int64_t* tokens, const float* emb, const float* pos, float* out, int B, int L, int Lp,
                             int D, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(tokens && emb && pos && out, "null pointer");
    DS_CHECK_ARG(B > 0 && L > 0 && Lp >= L && D % 4 == 0, "bad shape");
    const int M = B * Lp;
    hipLaunchKernelGGL(ds_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tokens, emb, pos, out, M, L, D, 0, Lp);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_adaln(const float* x, float* y, int M, int L, int D, const float* table, const int64_t* t,
                        ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && y && table && t, "null pointer");
    DS_CHECK_ARG(D == 1024, "only D = 1024 is built");
    hipLaunchKernelGGL((ds_layernorm_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, stream, x, y, M, L, 0, table,
                       t, (const float*)nullptr, (const float*)nullptr, 1e-5f, 0);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_layernorm(const float* x, float* y, int M, int D, const float* gamma, const float* beta,
                            ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && y && gamma && beta, "null pointer");
    DS_CHECK_ARG(D == 1024, "only D = 1024 is built");
    hipLaunchKernelGGL((ds_layernorm_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, stream, x, y, M, 1, 1,
                       (const float*)nullptr, (const int64_t*)nullptr, gamma, beta, 1e-5f, 0);
    DS_CHECK_LAUNCH();
    return 0;
}

// x: [B][P][C] channels-last.  work: >= B * ceil(P/256) * 2 * C doubles.  scale/shift: [B][C].
extern "C" int ds_groupnorm_stats(const float* x, int B, int P, int C, int groups, const float* gamma,
                                  const float* beta, float eps, double* work, float* scale, float* shift,
                                  ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && gamma && beta && work && scale && shift, "null pointer");
    DS_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0, "bad group count");
    DS_CHECK_ARG(C % 4 == 0 && C <= 1024 && ((uintptr_t)x & 15) == 0, "C % 4 == 0, C <= 1024, 16-byte aligned x");
    const int nchunk = (P + GN_PCHUNK - 1) / GN_PCHUNK;
    hipLaunchKernelGGL(ds_gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, stream, x, work, P, C);
    DS_CHECK_LAUNCH();
    hipLaunchKernelGGL(ds_gn_finish_kernel, dim3(groups, B), dim3(256), 0, stream, work, nchunk, P, C, groups, gamma, beta,
                       eps, scale, shift);
    DS_CHECK_LAUNCH();
    return 0;
}

// The second half alone: `part` = [B][nchunk][2][C] double partial sums (sum, sum of squares per channel) that another
// kernel produced -- ds_conv3x3_f16x2 writes them per output tile, so the statistics read of the tensor disappears.
extern "C" int ds_groupnorm_finish(const double* part, int B, int nchunk, int P, int C, int groups, const float* gamma,
                                   const float* beta, float eps, float* scale, float* shift, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(part && gamma && beta && scale && shift, "null pointer");
    DS_CHECK_ARG(B > 0 && nchunk > 0 && P > 0 && groups > 0 && groups <= 64 && C % groups == 0, "bad shape / group count");
    hipLaunchKernelGGL(ds_gn_finish_kernel, dim3(groups, B), dim3(256), 0, stream, part, nchunk, P, C, groups, gamma, beta, eps,
                       scale, shift);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- fp16-semantics variants for the CLIP text tower (clip/model.py:150-198, 341-354) -------------------
// The reference runs CLIP with fp16 weights and activations; here storage stays fp32 and every op's
// output is rounded to the fp16 grid, which reproduces those semantics up to accumulation order.
extern "C" int ds_embed_f16(const int64_t* tokens, const float* emb, const float* pos, float* out, int M, int L,
                            int D, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(tokens && emb && pos && out, "null pointer");
    DS_CHECK_ARG(M > 0 && L > 0 && D % 4 == 0, "bad shape");
    hipLaunchKernelGGL(ds_embed_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tokens, emb, pos, out, M, L, D, 1, L);
    DS_CHECK_LAUNCH();
    return 0;
}

// LayerNorm computed in fp32 on fp16-grid input, output rounded to fp16 (clip/model.py:150-157)
extern "C" int ds_layernorm_f16(const float* x, float* y, int M, int D, const float* gamma, const float* beta,
                                ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && y && gamma && beta, "null pointer");
    DS_CHECK_ARG(D == 512 || D == 1024, "D must be 512 or 1024");
    if (D == 512)
        hipLaunchKernelGGL((ds_layernorm_kernel<512>), dim3((M + 3) / 4), dim3(256), 0, stream, x, y, M, 1, 1,
                           (const float*)nullptr, (const int64_t*)nullptr, gamma, beta, 1e-5f, 1);
    else
        hipLaunchKernelGGL((ds_layernorm_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, stream, x, y, M, 1, 1,
                           (const float*)nullptr, (const int64_t*)nullptr, gamma, beta, 1e-5f, 1);
    DS_CHECK_LAUNCH();
    return 0;
}

// y = x / ||x||_2 per row with the norm and the quotient rounded to fp16
// (clip_text_embedding.py:79-80 on an fp16 tensor)
__global__ __launch_bounds__(256) void ds_l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int M,
                                                             int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float n = ds_r16(sqrtf(wave_sum(s)));
    float* yr = y + (size_t)row * D;
    for (int c = lane; c < D; c += 64) yr[c] = ds_r16(xr[c] / n);
}

extern "C" int ds_l2norm_rows_f16(const float* x, float* y, int M, int D, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && y && M > 0 && D > 0, "bad arguments");
    hipLaunchKernelGGL(ds_l2norm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, y, M, D);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- the same norms writing packed split planes for the f16x2 GEMM: yh = 2 planes of ceil16(M) * D halves ----
extern "C" int ds_adaln_split(const float* x, void* yh, int M, int L, int D, const float* table, const int64_t* t,
                              ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && yh && table && t, "null pointer");
    DS_CHECK_ARG(D == 1024, "only D = 1024 is built");
    hipLaunchKernelGGL((ds_layernorm_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, stream, x, (float*)yh, M, L, 0,
                       table, t, (const float*)nullptr, (const float*)nullptr, 1e-5f, 2);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_layernorm_split(const float* x, void* yh, int M, int D, const float* gamma, const float* beta,
                                  ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && yh && gamma && beta, "null pointer");
    DS_CHECK_ARG(D == 1024, "only D = 1024 is built");
    hipLaunchKernelGGL((ds_layernorm_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, stream, x, (float*)yh, M, 1, 1,
                       (const float*)nullptr, (const int64_t*)nullptr, gamma, beta, 1e-5f, 2);
    DS_CHECK_LAUNCH();
    return 0;
}
