// Small HBM/latency-bound kernels around the contractions.
#include "common.h"

// ---- codebook lookup fused with the ColumnMajor inverse permutation ---------------------------
// out[b][h][w][:] = E[tokens[b][w*H + h]][:]      (channels-last quant image)
// replaces first_stage_permuter(reverse=True) + get_codebook_entry's one-hot matmul
// (specvqgan/modules/transformer/permuter.py:31-55, vqvae/quantize.py:88-103).
__global__ __launch_bounds__(256) void ds_codebook_gather_kernel(const int64_t* __restrict__ tok,
                                                                 const float* __restrict__ E,
                                                                 float* __restrict__ out, int B, int H, int W,
                                                                 int C, int K) {
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= B * H * W) return;
    const int lane = threadIdx.x & 63;
    const int b = pix / (H * W), rem = pix - b * H * W;
    const int h = rem / W, w = rem - h * W;
    long long t = tok[(size_t)b * H * W + w * H + h];
    if (t < 0) t = 0;
    if (t >= K) t = K - 1;  // a leftover [MASK] cannot be decoded; the reference would raise
    const float* e = E + (size_t)t * C;
    float* o = out + (size_t)pix * C;
    for (int c = lane * 4; c < C; c += 256) *(f32x4*)(o + c) = *(const f32x4*)(e + c);
}

// ---- row softmax with scale, in place, zero-filling the padded tail -----------------------------
// x[row][0..n) <- softmax(scale * x[row][0..n)), x[row][n..ld) <- 0      (AttnBlock, model.py:214-216)
__global__ __launch_bounds__(256) void ds_softmax_rows_kernel(float* __restrict__ x, int rows, int n, int ld,
                                                              float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* xr = x + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, xr[c] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += expf(xr[c] * scale - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / s;
    for (int c = lane; c < ld; c += 64) xr[c] = c < n ? expf(xr[c] * scale - mx) * inv : 0.f;
}

// ---- single-output-channel convs: the per-tap dot products come from the GEMM (N = taps), these
//      kernels add the taps up with the conv's padding rule ---------------------------------------
// 3x3, zero padding 1 (Decoder.conv_out, model.py:633-637):  out[b][y][x] = bias + sum_tap T[b][y+ky-1][x+kx-1][tap]
__global__ __launch_bounds__(256) void ds_stencil9_kernel(const float* __restrict__ T, int ldt, float bias,
                                                          float* __restrict__ out, int B, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H * W) return;
    const int b = i / (H * W), rem = i - b * H * W;
    const int y = rem / W, x = rem - y * W;
    float acc = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sy = y + ky - 1, sx = x + kx - 1;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc += T[((size_t)(b * H + sy) * W + sx) * ldt + ky * 3 + kx];
        }
    out[i] = acc;
}

// k = 7, ReflectionPad1d(3), tanh (Generator tail, vocoder/modules.py:119-124)
__global__ __launch_bounds__(256) void ds_stencil7_tanh_kernel(const float* __restrict__ T, int ldt, float bias,
                                                               float* __restrict__ out, int B, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, t = i - b * N;
    float acc = bias;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int s = t + k - 3;
        if (s < 0) s = -s;
        if (s >= N) s = 2 * (N - 1) - s;
        acc += T[((size_t)b * N + s) * ldt + k];
    }
    out[i] = tanhf(acc);
}

// ---- mel [B][C][T] -> channels-last [B][T][Cpad] with y = a*x + b, zero channel padding ----------
__global__ __launch_bounds__(256) void ds_mel_to_cl_kernel(const float* __restrict__ mel, float* __restrict__ out,
                                                           int B, int C, int T, int Cpad, float a, float bb) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over B*T*Cpad, channel fastest
    if (i >= B * T * Cpad) return;
    const int c = i % Cpad, bt = i / Cpad;
    const int b = bt / T, t = bt - b * T;
    out[i] = c < C ? a * mel[((size_t)b * C + c) * T + t] + bb : 0.f;
}

// ---- nearest-code search of VectorQuantizer.forward (vqvae/quantize.py:46-53) --------------------------
// d[k] = (sum_c z[c]^2 + ee[k]) - 2 * ze[k] with ze = z E^T from the GEMM and ee[k] = sum_c E[k][c]^2 -- the
// reference's expression, same association; argmin with the first index winning ties (torch.argmin).
__global__ __launch_bounds__(256) void ds_vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ ze,
                                                           const float* __restrict__ ee, int64_t* __restrict__ idx,
                                                           float* __restrict__ dmin, int M, int C, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    float zz = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = z[(size_t)row * C + c];
        zz += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zz += __shfl_xor(zz, o);
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {   // ascending k within a lane keeps the first minimum
        const float d = (zz + ee[k]) - 2.f * ze[(size_t)row * K + k];
        if (d < best) { best = d; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        idx[row] = bi;
        if (dmin) dmin[row] = best;
    }
}

// ---- C ABI ----------------------------------------------------------------------------------------
extern "C" int ds_vq_argmin(const float* z, const float* ze, const float* ee, int64_t* idx, float* dmin, int M, int C,
                            int K, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(z && ze && ee && idx && M > 0 && C > 0 && K > 0, "bad arguments");
    hipLaunchKernelGGL(ds_vq_argmin_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, z, ze, ee, idx, dmin, M, C, K);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_codebook_gather(const int64_t* tokens, const float* codebook, float* out, int B, int H, int W,
                                  int C, int K, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(tokens && codebook && out, "null pointer");
    DS_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    hipLaunchKernelGGL(ds_codebook_gather_kernel, dim3((B * H * W + 3) / 4), dim3(256), 0, stream, tokens, codebook,
                       out, B, H, W, C, K);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_softmax_rows(float* x, int rows, int n, int ld, float scale, ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(x && rows > 0 && n > 0 && ld >= n, "bad arguments");
    hipLaunchKernelGGL(ds_softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, rows, n, ld, scale);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- Encoder.conv_in (diffusionmodules/model.py:423-427, :480): 3x3, zero padding 1, ONE input channel -> Cout channels, written
// channels-last.  9 multiply-adds per output value: as an implicit GEMM the contraction is 9 long (it ran zero-padded to the 32-wide
// channel granule of the conv kernel: 1.15 ms per 20 mels at 87 TF-eq, plus a 174 MB zero fill and a strided copy to build the padded
// input); here it is a store-bound pass over the output.  A workgroup takes one image row (its three input rows staged in LDS); a
// thread keeps the 4 x 9 weights of its four output channels in registers and walks the row's pixels; the 32 threads of a pixel
// write its 512 contiguous bytes (Cout = 128).  The GroupNorm that follows gets its statistics from this pass (per-segment partial sums), like from the 3x3 kernel.
#define C1_WMAX 2046   // widest image row (three zero-padded input rows of it live in LDS: 24 KB)
__global__ __launch_bounds__(256) void ds_conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                            int Cout, double* __restrict__ gn_part) {
    __shared__ float xs[3][C1_WMAX + 2];              // input rows y-1, y, y+1 with the zero padding in place
    __shared__ float red[2][256][4];
    const int tpp = Cout >> 2;                        // threads per pixel (4 output channels each)
    const int cg = threadIdx.x % tpp, slot = threadIdx.x / tpp, slots = 256 / tpp;
    const int b = blockIdx.y, y = blockIdx.x;         // one image row of one sample per workgroup: no index divisions
    const float* xb = x + (size_t)b * H * W;
    for (int i = threadIdx.x; i < 3 * (W + 2); i += 256) {
        const int r = i / (W + 2), c = i - r * (W + 2), sy = y + r - 1, sx = c - 1;
        xs[r][c] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? xb[(size_t)sy * W + sx] : 0.f;
    }
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[c][k] = w[(size_t)(cg * 4 + c) * 9 + k];
    const f32x4 bv = *(const f32x4*)(bias + cg * 4);
    float* ob = out + ((size_t)b * H + y) * W * Cout;
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    __syncthreads();
    for (int xx = slot; xx < W; xx += slots) {
        f32x4 o = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)                // taps in the conv's own order (ky major): an fp32 multiply-add chain
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float t = xs[ky][xx + kx];      // (the threads of a pixel read the same word: an LDS broadcast)
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] += wr[c][ky * 3 + kx] * t;
            }
        *(f32x4*)(ob + (size_t)xx * Cout + cg * 4) = o;
        s1 += o;
        s2 += o * o;
    }
    if (gn_part) {
        // GroupNorm statistics of the OUTPUT (the next op is a GroupNorm): per-channel sum and sum of squares of this row,
        // slots added in a fixed order -> part[b][row][2][Cout] doubles (the format ds_groupnorm_finish reads)
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[0][threadIdx.x][c] = s1[c]; red[1][threadIdx.x][c] = s2[c]; }
        __syncthreads();
        if (slot == 0) {
            double* pp = gn_part + (((size_t)b * H + y) * 2) * Cout + cg * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double a1 = 0.0, a2 = 0.0;
                for (int sl = 0; sl < slots; ++sl) { a1 += (double)red[0][sl * tpp + cg][c]; a2 += (double)red[1][sl * tpp + cg][c]; }
                pp[c] = a1;
                pp[Cout + c] = a2;
            }
        }
    }
}

// chunks of GroupNorm partial sums per sample that ds_conv3x3_c1 writes for an H x W image
extern "C" int ds_conv3x3_c1_chunks(int H, int W) { (void)W; return H; }      // one chunk per image row

// x: f32 [B][H][W] (the one input channel), w: [Cout][9] (= Conv2d.weight [Cout][1][3][3]), bias [Cout], out: f32 [B][H][W][Cout];
// gn_part (may be null): [B][ds_conv3x3_c1_chunks(H, W)][2][Cout] doubles
extern "C" int ds_conv3x3_c1(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cout,
                             double* gn_part, ds_stream_t stream_) {
    DS_CHECK_ARG(x && w && bias && out && B > 0 && H > 0 && W > 0, "bad arguments");
    DS_CHECK_ARG(Cout % 4 == 0 && Cout >= 4 && Cout <= 1024 && 1024 % Cout == 0, "Cout: a multiple of 4 that divides 1024");
    DS_CHECK_ARG((((uintptr_t)bias | (uintptr_t)out) & 15) == 0, "bias / out must be 16-byte aligned");
    DS_CHECK_ARG(W <= C1_WMAX && B <= 65535, "rows of at most 2046 pixels, at most 65535 samples");
    hipLaunchKernelGGL(ds_conv3x3_c1_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream_, x, w, bias, out, H, W, Cout, gn_part);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_stencil9(const float* taps, int ldt, float bias, float* out, int B, int H, int W,
                           ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(taps && out && ldt >= 9, "bad arguments");
    hipLaunchKernelGGL(ds_stencil9_kernel, dim3((B * H * W + 255) / 256), dim3(256), 0, stream, taps, ldt, bias, out,
                       B, H, W);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_stencil7_tanh(const float* taps, int ldt, float bias, float* out, int B, int N,
                                ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(taps && out && ldt >= 7, "bad arguments");
    hipLaunchKernelGGL(ds_stencil7_tanh_kernel, dim3((B * N + 255) / 256), dim3(256), 0, stream, taps, ldt, bias, out,
                       B, N);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_mel_to_cl(const float* mel, float* out, int B, int C, int T, int Cpad, float a, float b,
                            ds_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DS_CHECK_ARG(mel && out && Cpad >= C, "bad arguments");
    hipLaunchKernelGGL(ds_mel_to_cl_kernel, dim3((B * T * Cpad + 255) / 256), dim3(256), 0, stream, mel, out, B, C, T,
                       Cpad, a, b);
    DS_CHECK_LAUNCH();
    return 0;
}
