// MelGAN's HBM-bound layers as single-pass kernels (vocoder/modules.py:72-85 ResnetBlock at 32 / 64 channels, :104-113 the two
// stride-2 ConvTranspose1d, :119-124 final conv + tanh): ds_melgan_rb32_kernel, ds_melgan_rb64_kernel, ds_melgan_convt2_kernel,
// ds_melgan_final32_kernel.  Each reads its input tensor once and writes its output once.
//
// At 32 channels and 217 088 samples per clip a ResnetBlock is HBM-bound: its tensor is 1.78 GB per 64 clips, and the
// two-launch form (dilated k3 conv -> h, then [LReLU(h) | x] x [W2 | Ws]^T) moves it five times (x, h out, h in, x, y).
// ds_melgan_rb32_kernel moves it twice: a workgroup takes 128 consecutive time positions of one clip (+ dil halo rows each
// side, reflected at the clip's ends like ReflectionPad1d), stages LReLU(x) and x once as fp16 hi | lo planes in LDS, and
// runs both contractions TRANSPOSED -- H^T = W1 XL^T, Y^T = [W2 | Ws] [LReLU(H)^T ; X^T] -- so that a lane owns one time
// position throughout: the accumulator layout of the first product (lane = time, registers = channels 8 jj + 4 g + i) IS a
// B operand of the second once the k index of W2 is read in the same order, and LReLU(H) never leaves registers.  All
// weights (W1: 12, [W2 | Ws]: 8 fragments of 4 VGPRs) stay in registers of a persistent workgroup; the next tile's global
// loads are in flight while the current one is computed.  Same arithmetic as the two-launch form: 3-pass fp16 split
// (lo x hi, hi x lo, hi x hi per 16-k step, fp32 accumulate), h = acc * 2^-s1 + b1 in fp32, split again after LReLU.
//
// ds_melgan_final32_kernel: LReLU -> ReflectionPad1d(3) -> Conv1d(32 -> 1, k7) -> tanh in one pass over the 32-channel
// tensor (exact fp32 FMA chains per tap, taps added in order), instead of a 7-column fp32 GEMM + a stencil.
#include "common.h"

typedef _Float16 mg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 mg_h4 __attribute__((ext_vector_type(4)));

#define MG_TT 128          // time positions per tile
#define MG_PITCH 80        // bytes per LDS row of 32 halves (+ 16: the 16 rows a ds_read_b128 group touches hit 16 different slots)
#define MG_MAXDIL 16       // 5 float4 per thread cover (128 + 2 * 16) rows
#define MG_NLD 5
#define MG_YPITCH 144      // bytes per staged output row (32 floats + 4)

__device__ __forceinline__ float mg_lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

__device__ __forceinline__ void mg_split4(const f32x4& v, mg_h4& hi, mg_h4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = ds_split_hi(v[e]);
        lo[e] = ds_split_lo(v[e], hi[e]);
    }
}

__global__ __launch_bounds__(256, 2) void ds_melgan_rb32_kernel(const float* __restrict__ x, const _Float16* __restrict__ w3,
                                                                long long w3_plane, float s3, const float* __restrict__ b3,
                                                                const _Float16* __restrict__ wt, long long wt_plane, float st,
                                                                const float* __restrict__ bt, float* __restrict__ y, int T,
                                                                int dil, int tiles_per_clip, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tc = lane & 31, g = lane >> 5;
    const int R = MG_TT + 2 * dil;                       // staged rows of LReLU(x)
    unsigned char* XL = mg_smem;                         // [2 planes][R][80 B]
    const int PLXL = R * MG_PITCH;
    unsigned char* XS = mg_smem + 2 * PLXL;              // [4 waves][2 planes][32 rows][80 B]; a wave's block doubles as its output stage
    constexpr int XSW = 2 * 32 * MG_PITCH;               // 5120 B per wave (>= 32 * 144)

    // ---- weights and biases into registers, once ------------------------------------------------------------------
    mg_h8 a1h[3][2], a1l[3][2];                          // W1 fragments: rows = output channel tc', k = tap * 32 + 16 ks + 8 g ..
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const _Float16* p = w3 + (size_t)tc * 96 + tap * 32 + ks * 16 + g * 8;
            a1h[tap][ks] = *(const mg_h8*)p;
            a1l[tap][ks] = *(const mg_h8*)(p + w3_plane);
        }
    mg_h8 a2h[4], a2l[4];                                // [W2 | Ws] fragments; the h part in the accumulator's channel order
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const _Float16* p = wt + (size_t)tc * 64 + ks * 16 + g * 4;          // channels 16 ks + 4 g + {0..3} and + 8
        const mg_h4 h0 = *(const mg_h4*)p, h1 = *(const mg_h4*)(p + 8);
        const mg_h4 l0 = *(const mg_h4*)(p + wt_plane), l1 = *(const mg_h4*)(p + wt_plane + 8);
        a2h[ks] = mg_h8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        a2l[ks] = mg_h8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        const _Float16* q = wt + (size_t)tc * 64 + 32 + ks * 16 + g * 8;     // the shortcut's x channels, natural order
        a2h[2 + ks] = *(const mg_h8*)q;
        a2l[2 + ks] = *(const mg_h8*)(q + wt_plane);
    }
    float b3v[16], btv[16];                              // bias of the channel accumulator register j holds: 8 (j >> 2) + 4 g + (j & 3)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = 8 * (j >> 2) + 4 * g + (j & 3);
        b3v[j] = b3[n];
        btv[j] = bt[n];
    }

    f32x4 pre[MG_NLD];
    auto load_tile = [&](int tile) {
        const int b = tile / tiles_per_clip, t0 = (tile - b * tiles_per_clip) * MG_TT;
        const float* xb = x + (size_t)b * T * 32;
#pragma unroll
        for (int k = 0; k < MG_NLD; ++k) {
            const int i = tid + 256 * k, r = i >> 3, c4 = i & 7;
            if (r < R) {
                int t = t0 - dil + r;
                if (t < 0) t = -t;
                if (t >= T) t = 2 * (T - 1) - t;
                pre[k] = *(const f32x4*)(xb + (size_t)t * 32 + c4 * 4);
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int k = 0; k < MG_NLD; ++k) {
            const int i = tid + 256 * k, r = i >> 3, c4 = i & 7;
            if (r < R) {
                const f32x4 v = pre[k];
                f32x4 l;
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = mg_lrelu(v[e]);
                mg_h4 hi, lo;
                mg_split4(l, hi, lo);
                *(mg_h4*)(XL + r * MG_PITCH + c4 * 8) = hi;
                *(mg_h4*)(XL + PLXL + r * MG_PITCH + c4 * 8) = lo;
                const int rr = r - dil;
                if (rr >= 0 && rr < MG_TT) {
                    mg_split4(v, hi, lo);
                    unsigned char* d = XS + (rr >> 5) * XSW + (rr & 31) * MG_PITCH + c4 * 8;
                    *(mg_h4*)d = hi;
                    *(mg_h4*)(d + 32 * MG_PITCH) = lo;
                }
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        write_tile();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < n_tiles) load_tile(next);

        // ---- H^T = W1 x LReLU(X)^T over the three taps ------------------------------------------------------------
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned char* xl = XL + (wave * 32 + tc) * MG_PITCH + g * 16;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* p = xl + tap * dil * MG_PITCH + ks * 32;
                const mg_h8 bh = *(const mg_h8*)p, bl = *(const mg_h8*)(p + PLXL);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[tap][ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l[tap][ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[tap][ks], bh, acc, 0, 0, 0);
            }
        // ---- h = acc 2^-s1 + b1, LReLU, split: registers 8 ks .. 8 ks + 7 are the B fragment of k-step ks ------------
        mg_h8 hh[2], hl[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float hv = mg_lrelu(acc[j] * s3 + b3v[j]);
            const _Float16 hi = ds_split_hi(hv);
            hh[j >> 3][j & 7] = hi;
            hl[j >> 3][j & 7] = ds_split_lo(hv, hi);
        }
        // ---- Y^T = [W2 | Ws] x [LReLU(H)^T ; X^T] ----------------------------------------------------------------
        f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h[ks], hl[ks], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2l[ks], hh[ks], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h[ks], hh[ks], acc2, 0, 0, 0);
        }
        unsigned char* xs = XS + wave * XSW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* p = xs + tc * MG_PITCH + ks * 32 + g * 16;
            const mg_h8 bh = *(const mg_h8*)p, bl = *(const mg_h8*)(p + 32 * MG_PITCH);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h[2 + ks], bl, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2l[2 + ks], bh, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h[2 + ks], bh, acc2, 0, 0, 0);
        }
        // ---- y = acc2 2^-s2 + (b2 + bs): through the wave's own LDS block (its X^T rows are consumed), 128-byte rows out ----
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc2[4 * jj + e] * st + btv[4 * jj + e];
            *(f32x4*)(xs + tc * MG_YPITCH + (8 * jj + 4 * g) * 4) = o;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        {
            const int b = tile / tiles_per_clip, t0 = (tile - b * tiles_per_clip) * MG_TT;
            float* yb = y + ((size_t)b * T + t0 + wave * 32) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane + 64 * q, r = i >> 3, c4 = i & 7;
                *(f32x4*)(yb + r * 32 + c4 * 4) = *(const f32x4*)(xs + r * MG_YPITCH + c4 * 16);
            }
        }
        __syncthreads();
    }
}

// ---- 64 channels (the stage before: 108 544 samples per clip, the same 1.78 GB per tensor) ----------------------------------
// The same transposed two-product scheme with 256 time positions per tile and eight waves of 32 positions.  80 weight
// fragments no longer fit a wave's registers: both matrices sit in LDS for the lifetime of the persistent workgroup, stored
// FRAGMENT-MAJOR (one KB per (row block, k-step, plane): lane l's 16 bytes at l * 16, conflict-free and address-free), 80 KB;
// the LReLU(x) image takes the other 79 KB, so the shortcut's operand never enters LDS: every lane loads exactly the 8-channel
// chunks of ITS row that are its B fragments of x (and, after LReLU, its 16-byte pieces of the image), splits them in
// registers and keeps them there.  The output is staged through the image's space once all waves are past the first product.
#define MG64_TT 256
#define MG64_PITCH 144     // bytes per image row of 64 halves (+ 16)
#define MG64_MAXDIL 9
#define MG64_W1 0          // 48 fragments of W1:  ((nb * 3 + tap) * 4 + ks) * 2 + plane
#define MG64_W2 49152      // 32 fragments of [W2 | Ws]:  (mb * 8 + k2) * 2 + plane; k2 < 4: LReLU(h) channels in accumulator order
#define MG64_BIAS 81920    // b1 [64] | b2 + bs [64]
#define MG64_XL 82432
#define MG64_YPITCH 272    // bytes per staged output row (64 floats + 4)

__device__ __forceinline__ void mg_split8(const f32x4& a, const f32x4& b, mg_h8& hi, mg_h8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = ds_split_hi(a[e]);
        lo[e] = ds_split_lo(a[e], hi[e]);
        hi[4 + e] = ds_split_hi(b[e]);
        lo[4 + e] = ds_split_lo(b[e], hi[4 + e]);
    }
}

__global__ __launch_bounds__(512, 2) void ds_melgan_rb64_kernel(const float* __restrict__ x, const _Float16* __restrict__ w3,
                                                                long long w3_plane, float s3, const float* __restrict__ b3,
                                                                const _Float16* __restrict__ wt, long long wt_plane, float st,
                                                                const float* __restrict__ bt, float* __restrict__ y, int T,
                                                                int dil, int tiles_per_clip, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tc = lane & 31, g = lane >> 5;
    const int R = MG64_TT + 2 * dil;
    const int PLXL = R * MG64_PITCH;
    unsigned char* XL = mg_smem + MG64_XL;               // [2 planes][R][144 B]
    const float* bias_s = (const float*)(mg_smem + MG64_BIAS);

    // ---- weights into LDS, fragment-major, once ----------------------------------------------------------------------
    for (int f = wave; f < 48; f += 8) {
        const int plane = f & 1, ks = (f >> 1) & 3, tn = f >> 3, tap = tn % 3, nb = tn / 3;
        *(mg_h8*)(mg_smem + MG64_W1 + f * 1024 + lane * 16) =
            *(const mg_h8*)(w3 + plane * w3_plane + (size_t)(nb * 32 + tc) * 192 + tap * 64 + ks * 16 + g * 8);
    }
    for (int f = wave; f < 32; f += 8) {
        const int plane = f & 1, k2 = (f >> 1) & 7, mb = f >> 4;
        const _Float16* row = wt + plane * wt_plane + (size_t)(mb * 32 + tc) * 128;
        mg_h8 v;
        if (k2 < 4) {                                    // channels 16 k2 + 4 g + {0..3} and the same + 8
            const mg_h4 h0 = *(const mg_h4*)(row + k2 * 16 + g * 4), h1 = *(const mg_h4*)(row + k2 * 16 + g * 4 + 8);
            v = mg_h8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        } else {
            v = *(const mg_h8*)(row + 64 + (k2 - 4) * 16 + g * 8);
        }
        *(mg_h8*)(mg_smem + MG64_W2 + f * 1024 + lane * 16) = v;
    }
    if (tid < 64) ((float*)(mg_smem + MG64_BIAS))[tid] = b3[tid];
    else if (tid < 128) ((float*)(mg_smem + MG64_BIAS))[tid] = bt[tid - 64];

    f32x4 pre[8], preh;
    auto load_tile = [&](int tile) {
        const int b = tile / tiles_per_clip, t0 = (tile - b * tiles_per_clip) * MG64_TT;
        const float* xb = x + (size_t)b * T * 64;
        const float* own = xb + (size_t)(t0 + wave * 32 + tc) * 64 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            pre[2 * ks] = *(const f32x4*)(own + ks * 16);
            pre[2 * ks + 1] = *(const f32x4*)(own + ks * 16 + 4);
        }
        if (tid < 32 * dil) {                            // 2 dil halo rows of 16 float4
            const int hr = tid >> 4, c4 = tid & 15, r = hr < dil ? hr : MG64_TT + hr;
            int t = t0 - dil + r;
            if (t < 0) t = -t;
            if (t >= T) t = 2 * (T - 1) - t;
            preh = *(const f32x4*)(xb + (size_t)t * 64 + c4 * 4);
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        // ---- this tile's rows: x fragments stay in registers, LReLU(x) goes to the image -----------------------------------
        mg_h8 xsh[4], xsl[4];
        {
            unsigned char* d = XL + (dil + wave * 32 + tc) * MG64_PITCH + g * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mg_split8(pre[2 * ks], pre[2 * ks + 1], xsh[ks], xsl[ks]);
                f32x4 l0, l1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    l0[e] = mg_lrelu(pre[2 * ks][e]);
                    l1[e] = mg_lrelu(pre[2 * ks + 1][e]);
                }
                mg_h8 hi, lo;
                mg_split8(l0, l1, hi, lo);
                *(mg_h8*)(d + ks * 32) = hi;
                *(mg_h8*)(d + ks * 32 + PLXL) = lo;
            }
            if (tid < 32 * dil) {
                const int hr = tid >> 4, c4 = tid & 15, r = hr < dil ? hr : MG64_TT + hr;
                f32x4 l;
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = mg_lrelu(preh[e]);
                mg_h4 hi, lo;
                mg_split4(l, hi, lo);
                *(mg_h4*)(XL + r * MG64_PITCH + c4 * 8) = hi;
                *(mg_h4*)(XL + PLXL + r * MG64_PITCH + c4 * 8) = lo;
            }
        }
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < n_tiles) load_tile(next);

        // ---- H^T = W1 x LReLU(X)^T --------------------------------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[nb][j] = 0.f;
        {
            const unsigned char* xl = XL + (wave * 32 + tc) * MG64_PITCH + g * 16;
            const unsigned char* wl = mg_smem + MG64_W1 + lane * 16;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const unsigned char* p = xl + tap * dil * MG64_PITCH + ks * 32;
                    const mg_h8 bh = *(const mg_h8*)p, bl = *(const mg_h8*)(p + PLXL);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const unsigned char* a = wl + (((nb * 3 + tap) * 4 + ks) * 2) * 1024;
                        const mg_h8 ah = *(const mg_h8*)a, al = *(const mg_h8*)(a + 1024);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nb], 0, 0, 0);
                    }
                }
        }
        __syncthreads();                                 // every wave is past its image reads: the image's space takes the output

        // ---- h = acc 2^-s1 + b1, LReLU, split: accumulator registers 8 k' .. 8 k' + 7 of block nb = k-step 2 nb + k' ----------
        mg_h8 hh[4], hl[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 bv = *(const f32x4*)(bias_s + nb * 32 + jj * 8 + g * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = jj * 4 + e;
                    const float hv = mg_lrelu(acc[nb][j] * s3 + bv[e]);
                    const _Float16 hi = ds_split_hi(hv);
                    hh[nb * 2 + (j >> 3)][j & 7] = hi;
                    hl[nb * 2 + (j >> 3)][j & 7] = ds_split_lo(hv, hi);
                }
            }
        // ---- Y^T = [W2 | Ws] x [LReLU(H)^T ; X^T] ------------------------------------------------------------------------
        f32x16 acc2[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[mb][j] = 0.f;
        {
            const unsigned char* wl = mg_smem + MG64_W2 + lane * 16;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const mg_h8 bh = k2 < 4 ? hh[k2 & 3] : xsh[k2 & 3], bl = k2 < 4 ? hl[k2 & 3] : xsl[k2 & 3];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const unsigned char* a = wl + ((mb * 8 + k2) * 2) * 1024;
                    const mg_h8 ah = *(const mg_h8*)a, al = *(const mg_h8*)(a + 1024);
                    acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[mb], 0, 0, 0);
                    acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[mb], 0, 0, 0);
                    acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2[mb], 0, 0, 0);
                }
            }
        }
        // ---- y = acc2 2^-s2 + (b2 + bs): the wave's 32 rows through LDS, 256-byte rows out ---------------------------------
        unsigned char* ys = XL + wave * (32 * MG64_YPITCH);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 bv = *(const f32x4*)(bias_s + 64 + mb * 32 + jj * 8 + g * 4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc2[mb][4 * jj + e] * st + bv[e];
                *(f32x4*)(ys + tc * MG64_YPITCH + (mb * 32 + 8 * jj + 4 * g) * 4) = o;
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        {
            const int b = tile / tiles_per_clip, t0 = (tile - b * tiles_per_clip) * MG64_TT;
            float* yb = y + ((size_t)b * T + t0 + wave * 32) * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = lane + 64 * q, r = i >> 4, c4 = i & 15;
                *(f32x4*)(yb + r * 64 + c4 * 4) = *(const f32x4*)(ys + r * MG64_YPITCH + c4 * 16);
            }
        }
        __syncthreads();
    }
}

// Whether the single-pass kernel takes this block: 32 channels, whole 128-position tiles, the halo within the staging loop.
extern "C" int ds_melgan_resblock_fused_ok(int T, int C, int dil) {
    if (C == 32) return T > 0 && T % MG_TT == 0 && dil > 0 && dil <= MG_MAXDIL && dil < T;
    if (C == 64) return T > 0 && T % MG64_TT == 0 && dil > 0 && dil <= MG64_MAXDIL && dil < T;
    return 0;
}

int ds_launch_melgan_rb(const float* x, const void* w3, long long w3_plane, float w3_scale, const float* b3, const void* wt,
                        long long wt_plane, float wt_scale, const float* bt, float* y, int B, int T, int C, int dil, hipStream_t s) {
    static int wgs_per_cu_dev[64], n_cu_dev[64];      // per device (DsOnce rb32_once below)
    static DsOnce rb32_once;
    if (C == 64) {                                       // one 8-wave workgroup per CU (161 KB of LDS)
        static DsOnce attr_set;
        const size_t lds64 = (size_t)MG64_XL + 2 * (size_t)(MG64_TT + 2 * dil) * MG64_PITCH;
        if (attr_set.need()) {
            if (hipFuncSetAttribute((const void*)ds_melgan_rb64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    MG64_XL + 2 * (MG64_TT + 2 * MG64_MAXDIL) * MG64_PITCH) != hipSuccess) {
                ds_set_error("ds_melgan_resblock: cannot reserve %d bytes of LDS", MG64_XL + 2 * (MG64_TT + 2 * MG64_MAXDIL) * MG64_PITCH);
                return -1;
            }
            attr_set.done();
        }
        int dev = 0, cus = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int tpc = T / MG64_TT;
        const long long nt = (long long)B * tpc;
        DS_CHECK_ARG(nt < (1ll << 31), "too many tiles");
        const long long grid64 = nt < cus ? nt : cus;
        hipLaunchKernelGGL(ds_melgan_rb64_kernel, dim3((unsigned)grid64), dim3(512), lds64, s, x, (const _Float16*)w3, w3_plane,
                           w3_scale, b3, (const _Float16*)wt, wt_plane, wt_scale, bt, y, T, dil, tpc, (int)nt);
        DS_CHECK_LAUNCH();
        return 0;
    }
    const int R = MG_TT + 2 * dil;
    const size_t lds = (size_t)2 * R * MG_PITCH + 4 * 2 * 32 * MG_PITCH;
    const int devi = DsOnce::dev();
    if (rb32_once.need()) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, dev);
        n_cu_dev[devi] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ds_melgan_rb32_kernel, 256,
                                                         (size_t)2 * (MG_TT + 2 * MG_MAXDIL) * MG_PITCH + 8 * 32 * MG_PITCH) != hipSuccess || occ < 1)
            occ = 2;
        wgs_per_cu_dev[devi] = occ;
        rb32_once.done();
    }
    const int n_cu = n_cu_dev[devi], wgs_per_cu = wgs_per_cu_dev[devi];
    const int tiles_per_clip = T / MG_TT;
    const long long n_tiles = (long long)B * tiles_per_clip;
    DS_CHECK_ARG(n_tiles < (1ll << 31), "too many tiles");
    long long grid = (long long)n_cu * wgs_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(ds_melgan_rb32_kernel, dim3((unsigned)grid), dim3(256), lds, s, x, (const _Float16*)w3, w3_plane, w3_scale,
                       b3, (const _Float16*)wt, wt_plane, wt_scale, bt, y, T, dil, tiles_per_clip, (int)n_tiles);
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- ConvTranspose1d(k = 4, stride 2, padding 1) of the two last upsampling stages (128 -> 64, 64 -> 32 channels) ---------------
// LeakyReLU in front (vocoder/modules.py:104-113).  out[2 q + 1] = W[:, :, 0] a[q + 1] + W[:, :, 2] a[q],  out[2 q] = W[:, :, 1] a[q]
// + W[:, :, 3] a[q - 1]  (a = LReLU(x), rows outside [0, Tin) are zero): two phases over the same input rows.  As r polyphase
// GEMMs on the gather kernel each phase re-read and re-split the input (2.2 x the layer's byte floor).  Here a workgroup stages the
// LReLU(x) rows of 128 input positions (+ 1 each side) once as fp16 hi | lo planes, eight waves = (phase, 32-channel block,
// position range) each keep THEIR weight fragments in registers for the lifetime of the persistent workgroup (transposed product:
// lane = position), and the 256 output rows of the tile -- one contiguous block of y -- leave through LDS as full rows.
template <int CIN, int COUT>
__global__ __launch_bounds__(512, 2) void ds_melgan_convt2_kernel(const float* __restrict__ x, const _Float16* __restrict__ w,
                                                                  long long w_plane, float osc, const float* __restrict__ bias,
                                                                  float* __restrict__ y, int Tin, int tiles_per_clip, int n_tiles) {
    constexpr int NMB = COUT / 32, NCOMBO = 2 * NMB, NSPLIT = 8 / NCOMBO, NPOS = 128 / NSPLIT, NCB = NPOS / 32;
    constexpr int KS = CIN / 16;                         // k-steps per tap
    constexpr int PITCH = CIN * 2 + 16;                  // image row bytes
    constexpr int ROWS = 130;
    constexpr int PL = ROWS * PITCH;
    constexpr int YP = COUT * 4 + 16;                    // staged output row bytes
    constexpr int NLD = (ROWS * CIN / 4 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_smem[];
    unsigned char* IM = mg_smem;                         // [2 planes][130][PITCH]
    unsigned char* YS = mg_smem + 2 * PL;                // [256][YP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tc = lane & 31, g = lane >> 5;
    const int combo = wave / NSPLIT, part = wave % NSPLIT, ph = combo / NMB, mb = combo % NMB;
    const int e_p = ph == 0 ? 1 : 0;                     // source index s0 = q + e_p; taps read s0 and s0 - 1

    mg_h8 ah[2][KS], al[2][KS];                          // this wave's weights: rows ph * COUT + mb * 32 + tc, k = tap * CIN + 16 ks + 8 g ..
#pragma unroll
    for (int tap = 0; tap < 2; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const _Float16* p = w + (size_t)(ph * COUT + mb * 32 + tc) * (2 * CIN) + tap * CIN + ks * 16 + g * 8;
            ah[tap][ks] = *(const mg_h8*)p;
            al[tap][ks] = *(const mg_h8*)(p + w_plane);
        }
    float* BS = (float*)(mg_smem + 2 * PL + 256 * YP);   // the bias [COUT]
    if (tid < COUT) BS[tid] = bias[tid];

    f32x4 pre[NLD];
    auto load_tile = [&](int tile) {
        const int b = tile / tiles_per_clip, q0 = (tile - b * tiles_per_clip) * 128;
        const float* xb = x + (size_t)b * Tin * CIN;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                  // opaque: hipcc must not keep the nine (tile-invariant) row / column pairs in registers
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid_o + 512 * k, r = i / (CIN / 4), c4 = i - r * (CIN / 4);
            const int sidx = q0 - 1 + r;
            pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < ROWS && sidx >= 0 && sidx < Tin) pre[k] = *(const f32x4*)(xb + (size_t)sidx * CIN + c4 * 4);
        }
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid_w + 512 * k, r = i / (CIN / 4), c4 = i - r * (CIN / 4);
            if (r < ROWS) {
                f32x4 l;
#pragma unroll
                for (int e = 0; e < 4; ++e) l[e] = mg_lrelu(pre[k][e]);
                mg_h4 hi, lo;
                mg_split4(l, hi, lo);
                *(mg_h4*)(IM + r * PITCH + c4 * 8) = hi;
                *(mg_h4*)(IM + PL + r * PITCH + c4 * 8) = lo;
            }
        }
        __syncthreads();                                 // image written (and every wave is past the previous tile's output rows)
        const int next = tile + gridDim.x;
        if (next < n_tiles) load_tile(next);

        // one 32-position block at a time (16 accumulator registers next to the 4 CIN resident weight registers)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int j = part * NPOS + cb * 32 + tc;                         // position inside the tile; image row j + 1 + e_p - tap
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 2; ++tap)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned char* p = IM + (j + 1 + e_p - tap) * PITCH + ks * 32 + g * 16;
                    const mg_h8 bh = *(const mg_h8*)p, bl = *(const mg_h8*)(p + PL);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap][ks], bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap][ks], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap][ks], bh, acc, 0, 0, 0);
                    if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // (hipcc would hoist all 4 KS fragment reads: spills)
                }
            // output row of position j, phase ph: 2 j + (ph == 0); the wave's 32 channels as four 16-byte pieces per lane
            unsigned char* d = YS + (2 * j + e_p) * YP + (mb * 32 + 4 * g) * 4;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b4 = *(const f32x4*)(BS + mb * 32 + 8 * jj + 4 * g);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[4 * jj + e] * osc + b4[e];
                *(f32x4*)(d + jj * 32) = o;
            }
        }
        __syncthreads();                                 // rows complete; the image is free for the next tile
        {
            const int b = tile / tiles_per_clip, q0 = (tile - b * tiles_per_clip) * 128;
            float* yb = y + ((size_t)b * Tin * 2 + 2 * q0) * COUT;
            constexpr int PER_ROW = COUT / 4;
#pragma unroll
            for (int k = 0; k < 256 * PER_ROW / 512; ++k) {
                const int i = tid + 512 * k, r = i / PER_ROW, c4 = i - r * PER_ROW;
                if (2 * q0 + r < 2 * Tin) *(f32x4*)(yb + (size_t)r * COUT + c4 * 4) = *(const f32x4*)(YS + r * YP + c4 * 16);
            }
        }
    }
}

template <int CIN, int COUT>
static int mg_launch_convt2(const float* x, const void* w, long long w_plane, float osc, const float* bias, float* y, int B, int Tin,
                            hipStream_t s) {
    constexpr int lds = 2 * 130 * (CIN * 2 + 16) + 256 * (COUT * 4 + 16) + COUT * 4;
    static DsOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute((const void*)ds_melgan_convt2_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            ds_set_error("ds_melgan_convt2: cannot reserve %d bytes of LDS", lds);
            return -1;
        }
        attr_set.done();
    }
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int tpc = (Tin + 127) / 128;
    const long long nt = (long long)B * tpc;
    DS_CHECK_ARG(nt < (1ll << 31), "too many tiles");
    const long long grid = nt < cus ? nt : cus;
    hipLaunchKernelGGL((ds_melgan_convt2_kernel<CIN, COUT>), dim3((unsigned)grid), dim3(512), lds, s, x, (const _Float16*)w, w_plane, osc,
                       bias, y, Tin, tpc, (int)nt);
    DS_CHECK_LAUNCH();
    return 0;
}

// y [B][2 Tin][Cout] = ConvTranspose1d(k = 4, s = 2, p = 1)(LeakyReLU_0.2(x [B][Tin][Cin])), channels-last fp32.  w = the fp16
// planes of the polyphase weights * 2^s, [2 phases][Cout][2 taps][Cin] (phase p: taps W[:, :, p] on x[s0], W[:, :, p + 2] on
// x[s0 - 1]; what modeling/vocoder.py packs for the polyphase GEMMs), w_plane halves apart; out_scale = 2^-s.
// (Cin, Cout) = (128, 64) or (64, 32) are built.
extern "C" int ds_melgan_convt2(const float* x, const void* w, long long w_plane, float out_scale, const float* bias, float* y, int B,
                                int Tin, int Cin, int Cout, ds_stream_t stream) {
    DS_CHECK_ARG(x && w && bias && y && B > 0 && Tin > 1 && out_scale > 0.f, "bad arguments");
    DS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "operands must be 16-byte aligned");
    if (Cin == 128 && Cout == 64) return mg_launch_convt2<128, 64>(x, w, w_plane, out_scale, bias, y, B, Tin, (hipStream_t)stream);
    if (Cin == 64 && Cout == 32) return mg_launch_convt2<64, 32>(x, w, w_plane, out_scale, bias, y, B, Tin, (hipStream_t)stream);
    ds_set_error("ds_melgan_convt2: (Cin, Cout) = (%d, %d) is not built (128 -> 64 and 64 -> 32 are)", Cin, Cout);
    return -1;
}
extern "C" int ds_melgan_convt2_ok(int Cin, int Cout) { return (Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 32); }

// ---- final layer: out[b][t] = tanh(bias + sum_j sum_c w[j][c] LReLU(x[b][reflect(t + j - 3)][c])) ----------------------------
#define MG_FT 256          // outputs per workgroup
#define MG_FPITCH 36       // floats per staged row (32 + 4: conflict-free 16-byte reads down a column of rows)

__global__ __launch_bounds__(256) void ds_melgan_final32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float bias, float* __restrict__ out, int T, int tiles_per_clip) {
    __shared__ __attribute__((aligned(16))) float xs[(MG_FT + 6) * MG_FPITCH];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_per_clip, t0 = (blockIdx.x - b * tiles_per_clip) * MG_FT;
    const float* xb = x + (size_t)b * T * 32;
    for (int i = tid; i < (MG_FT + 6) * 8; i += 256) {
        const int r = i >> 3, c4 = i & 7;
        int t = t0 - 3 + r;
        if (t < 0) t = -t;
        if (t >= T) t = 2 * (T - 1) - t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *(const f32x4*)(xb + (size_t)t * 32 + c4 * 4);      // (rows past a ragged last tile: unused)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mg_lrelu(v[e]);
        *(f32x4*)(xs + r * MG_FPITCH + c4 * 4) = v;
    }
    __syncthreads();
    if (t0 + tid >= T) return;
    float acc = bias;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float* row = xs + (tid + j) * MG_FPITCH;
        float s = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const f32x4 v = *(const f32x4*)(row + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], w[j * 32 + c4 * 4 + e], s);      // uniform address: scalar loads
        }
        acc += s;
    }
    out[(size_t)b * T + t0 + tid] = tanhf(acc);
}

// x [B][T][32] channels-last fp32, w [7 taps][32] fp32 (the folded weight of Conv1d(32, 1, 7), tap-major), out [B][T].
extern "C" int ds_melgan_final(const float* x, const float* w, float bias, float* out, int B, int T, int C, ds_stream_t stream) {
    DS_CHECK_ARG(x && w && out && B > 0 && T > 3, "bad arguments");
    DS_CHECK_ARG(C == 32, "C = 32 is built (ngf = 32, the reference configuration)");
    const int tiles_per_clip = (T + MG_FT - 1) / MG_FT;
    hipLaunchKernelGGL(ds_melgan_final32_kernel, dim3((unsigned)(B * tiles_per_clip)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       out, T, tiles_per_clip);
    DS_CHECK_LAUNCH();
    return 0;
}
