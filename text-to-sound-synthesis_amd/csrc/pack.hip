// Operand preparation for the training step's split GEMMs, round 5 (modeling/train.py, precision "f16x2"):
//
//   ds_pack_operand   ONE pass over an fp32 matrix X[rows][cols] that writes everything the step's three GEMMs per linear
//                     layer (y = x W^T, dX = dY W, dW = dY^T X; engine/solver_spec.py:308-331 back-propagates exactly these)
//                     need of it, so that all of them run on packed split planes staged by LDS-DMA (gemm_f16x2.hip AMODE 2):
//                       * the ROW form   -- packed planes of X   (A operand of y = x W^T / dX = dY W;  W operand: W itself)
//                       * the TRANSPOSED form -- packed planes of X^T with the contraction index (the rows of X) zero-padded
//                         to rows_pad (A / W operands of dW = dY^T X;  W operand of dX: W^T); optionally as the k-range
//                         [t_col0, t_col0 + rows_pad) of a WIDER destination (t_cols): the query | key | value weights of a
//                         fused projection are packed straight into their row / k ranges of the fused operand -- no
//                         concatenated fp32 copy of the weights per step
//                       * per-tile-row column sums (bias gradients: db = column sums of dY), summed by ds_colsum afterwards
//                         in a fixed order -- no atomics, the gradients stay bit-reproducible; taken BEFORE `scale` is applied
//                         (round 6: `scale` of a gradient is its site's own power of two, which the bias gradient must not carry)
//                       * max |x * scale| (the loss-scale calibration / saturation monitor of the step)
//                     with an optional elementwise prologue, so that the MLP's activation never exists in fp32:
//                       DS_PACK_GELU2      x := gelu2(x)                  (forward: fc2's input from fc1's output)
//                       DS_PACK_GELU2_BWD  x := x * gelu2'(aux)           (backward: d fc1-output from d gelu-output)
//                     It replaces the round-2 ds_convert_operand (two transposing passes per linear layer and backward GEMM), the bias
//                     ds_colsum_ws launches and the ds_amax probes of rounds 2-4.
//   ds_adamw_multi    the AdamW update of 64 parameter tensors per launch (descriptors by value in the kernel arguments),
//                     16-byte accesses -- rounds 2-4 launched ds_adamw_dev once per tensor (462 launches per iteration).
//
// Packed planes: common.h ds_packed_off -- [ceil(R/16)][K/32] tiles of 16 rows x 32 k = 1 KB, 16-byte chunks swizzled.
#include "common.h"

typedef _Float16 pk_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pk_gelu2(float v) { return v / (1.f + expf(-1.702f * v)); }   // = ds_gelu2_kernel (train.hip)
__device__ __forceinline__ float pk_gelu2_grad(float v) {
    const float sg = 1.f / (1.f + expf(-1.702f * v));
    return sg + 1.702f * v * sg * (1.f - sg);
}

// 64 x 64 source tile per workgroup through LDS (rows padded to 65 words).  grid = (row tiles over max(rows_pad,
// ceil16(rows)), column tiles).  Source elements outside [rows) x [cols) read as zero, so the padding rows / k-slots of
// both forms are written as zeros.  Every element is split ONCE, when it is loaded (hi | lo << 16 in one LDS word, the
// staging format of the GEMM epilogues); the two forms are byte shuffles of those words (v_perm), the column sums come from
// the fp32 values still in registers (per-thread partials over its 4 rows, added across the 16 row groups in a fixed order).
template <int PRO>
__global__ __launch_bounds__(256) void ds_pack_operand_kernel(const float* __restrict__ src, int rows, int cols, long long ld_src,
                                                              float scale, const float* __restrict__ aux, long long ld_aux,
                                                              _Float16* __restrict__ dst_row, long long plane_row,
                                                              _Float16* __restrict__ dst_t, long long plane_t, int rows_pad,
                                                              int t_col0, int t_cols,
                                                              float* __restrict__ colsum_part, unsigned* __restrict__ amax) {
    __shared__ unsigned t[64][65];
    __shared__ float cs[16][64];
    __shared__ float wm[4];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    float m = 0.f;
    {
        const int c4 = (tid & 15) * 4, rq = tid >> 4;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = rq + 16 * it;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + r < rows && c0 + c4 < cols) {          // cols % 4 == 0: a 4-column unit is inside or outside
                v = *(const f32x4*)(src + (size_t)(r0 + r) * ld_src + c0 + c4);
                if (PRO == DS_PACK_GELU2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pk_gelu2(v[e]);
                } else if (PRO == DS_PACK_GELU2_BWD) {
                    const f32x4 u = *(const f32x4*)(aux + (size_t)(r0 + r) * ld_aux + c0 + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * pk_gelu2_grad(u[e]);
                }
                sum += v;                                   // column sums: of the values BEFORE `scale` (see ds_pack_operand)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] *= scale;
                    const float a = fabsf(v[e]);
                    m = a > m ? a : m;                      // NaN never wins (a calibration quantity, not a validity check)
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[r][c4 + e] = ds_split_pack(v[e]);
        }
        if (colsum_part) {
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[rq][c4 + e] = sum[e];
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        if ((tid & 63) == 0) wm[tid >> 6] = m;
    }
    __syncthreads();
    if (amax && tid == 0) {
        float b = wm[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = wm[w] > b ? wm[w] : b;
        if (b > 0.f) atomicMax(amax, __float_as_uint(b));
    }
    // column sums of this tile row: the 16 row-group partials in a fixed order -> partial[blockIdx.x][c]
    if (colsum_part && tid < 64 && c0 + tid < cols && r0 < rows) {
        float s_ = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s_ += cs[q][tid];
        colsum_part[(size_t)blockIdx.x * cols + c0 + tid] = s_;
    }
    // 8 packed words -> the 8 halves of plane 0 (low halves) and of plane 1 (high halves): one v_perm_b32 per output word
#define PK_UNZIP(x_, hi_, lo_)                                                                   \
    do {                                                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
            hi_[e] = __builtin_amdgcn_perm((x_)[2 * e + 1], (x_)[2 * e], 0x05040100u);           \
            lo_[e] = __builtin_amdgcn_perm((x_)[2 * e + 1], (x_)[2 * e], 0x07060302u);           \
        }                                                                                        \
    } while (0)
    typedef unsigned pk_u4 __attribute__((ext_vector_type(4)));
    // ROW form: 4 row groups x 2 k-tiles; one 16-byte chunk (8 consecutive columns of a row) per thread and iteration
    if (dst_row && r0 < ((rows + 15) & ~15)) {
        const int ktiles = cols >> 5;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int w = tid + 256 * it;                    // 512 chunks: row (64) x chunk (8)
            const int ch = w & 7, r = w >> 3;
            const int row = r0 + r, col = c0 + ch * 8;
            if (row < ((rows + 15) & ~15) && col < cols) {
                unsigned x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = t[r][ch * 8 + e];
                pk_u4 hi, lo;
                PK_UNZIP(x, hi, lo);
                _Float16* d = dst_row + ds_packed_off(row, col, ktiles);
                *(pk_u4*)d = hi;
                *(pk_u4*)(d + plane_row) = lo;
            }
        }
    }
    // TRANSPOSED form: logical X^T[cols][t_cols]; a chunk = 8 consecutive source rows of one source column
    if (dst_t && r0 < rows_pad) {
        const int ktiles = t_cols >> 5;                      // the destination's full contraction length (>= t_col0 + rows_pad)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int w = tid + 256 * it;                    // 512 chunks: source column (64) x row chunk (8)
            const int ch = w & 7, cl = w >> 3;
            const int trow = c0 + cl, tcol = r0 + ch * 8;    // row / column of X^T
            if (trow < ((cols + 15) & ~15) && tcol < rows_pad) {
                unsigned x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = t[ch * 8 + e][cl];
                pk_u4 hi, lo;
                PK_UNZIP(x, hi, lo);
                _Float16* d = dst_t + ds_packed_off(trow, t_col0 + tcol, ktiles);
                *(pk_u4*)d = hi;
                *(pk_u4*)(d + plane_t) = lo;
            }
        }
    }
#undef PK_UNZIP
}

extern "C" int ds_pack_operand_tile_rows(int rows, int rows_pad) {
    const int r16 = (rows + 15) & ~15;
    return ((rows_pad > r16 ? rows_pad : r16) + 63) / 64;
}

extern "C" int ds_pack_operand(const float* src, int rows, int cols, long long ld_src, float scale, int pro, const float* aux,
                               long long ld_aux, void* dst_row, long long plane_row, void* dst_t, long long plane_t,
                               int rows_pad, int t_col0, int t_cols, float* colsum_part, float* amax, ds_stream_t stream) {
    DS_CHECK_ARG(src && rows > 0 && cols > 0 && cols % 32 == 0 && ld_src >= cols && ld_src % 4 == 0, "src: cols % 32 == 0, ld % 4 == 0");
    DS_CHECK_ARG((((uintptr_t)src) & 15) == 0, "src must be 16-byte aligned");
    DS_CHECK_ARG(dst_row || dst_t || colsum_part || amax, "nothing to produce");
    DS_CHECK_ARG(pro == DS_PACK_PLAIN || pro == DS_PACK_GELU2 ||
                     (pro == DS_PACK_GELU2_BWD && aux && ld_aux >= cols && ld_aux % 4 == 0 && (((uintptr_t)aux) & 15) == 0),
                 "prologue: PLAIN, GELU2, or GELU2_BWD with a 16-byte aligned aux matrix");
    DS_CHECK_ARG(!dst_row || ((((uintptr_t)dst_row) & 15) == 0 && plane_row % 8 == 0 &&
                              plane_row >= (long long)((rows + 15) & ~15) * cols),
                 "row form: 16-byte aligned, plane stride >= ceil16(rows) * cols halves");
    if (t_cols == 0) { t_cols = rows_pad; t_col0 = 0; }      // the matrix is the whole destination
    DS_CHECK_ARG(!dst_t || ((((uintptr_t)dst_t) & 15) == 0 && rows_pad >= rows && rows_pad % 32 == 0 && plane_t % 8 == 0 &&
                            t_col0 >= 0 && t_col0 % 32 == 0 && t_cols % 32 == 0 && t_col0 + rows_pad <= t_cols &&
                            plane_t >= (long long)((cols + 15) & ~15) * t_cols),
                 "transposed form: rows_pad % 32 == 0 and >= rows, k-range [t_col0, t_col0 + rows_pad) inside t_cols, plane stride "
                 ">= ceil16(cols) * t_cols halves");
    if (!dst_t) rows_pad = 0;
    const dim3 grid((unsigned)ds_pack_operand_tile_rows(rows, rows_pad), (unsigned)((cols + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
#define PK_LAUNCH(P)                                                                                                       \
    hipLaunchKernelGGL(ds_pack_operand_kernel<P>, grid, dim3(256), 0, s, src, rows, cols, ld_src, scale, aux, ld_aux,      \
                       (_Float16*)dst_row, plane_row, (_Float16*)dst_t, plane_t, rows_pad, t_col0, t_cols, colsum_part, (unsigned*)amax)
    if (pro == DS_PACK_GELU2) PK_LAUNCH(DS_PACK_GELU2);
    else if (pro == DS_PACK_GELU2_BWD) PK_LAUNCH(DS_PACK_GELU2_BWD);
    else PK_LAUNCH(DS_PACK_PLAIN);
#undef PK_LAUNCH
    DS_CHECK_LAUNCH();
    return 0;
}

// ---- multi-tensor AdamW ----------------------------------------------------------------------------------------------
// Up to AW_BATCH tensor descriptors travel BY VALUE in the kernel arguments (3 KB of the 4 KB argument space): nothing is
// copied to the device beforehand, so the launch is captured into a hipGraph like any other (the training iteration replays
// as one graph) -- 8 launches for the denoiser's 462 parameter tensors instead of 462.  Tensor i of a batch owns the
// 4096-element chunks [first_chunk, first_chunk of i+1); a workgroup updates one chunk (binary search of its tensor),
// 16-byte accesses where the tensor's pointers allow.  hyper = { lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale } in
// device memory, as for ds_adamw_dev; the arithmetic is that kernel's, expression for expression.
struct DsAdamwTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n, first_chunk;
};
#define AW_CHUNK 4096
#define AW_BATCH 64
struct DsAdamwBatch {
    DsAdamwTensor t[AW_BATCH];
    int n;
};

__device__ __forceinline__ void aw_update(float& p, float g, float& m, float& v, float lr, float bc1, float bc2s, float gs, float b1,
                                          float b2, float eps, float wd) {
    const float gi = g * gs;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    float pi = p * (1.f - lr * wd);
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    p = pi;
}

__global__ __launch_bounds__(256) void ds_adamw_multi_kernel(const DsAdamwBatch batch, const float* __restrict__ hyper, float b1,
                                                             float b2, float eps, float wd) {
    const long long chunk = blockIdx.x;
    int lo = 0, hi = batch.n - 1;
    while (lo < hi) {                                // last tensor whose first_chunk <= chunk (block-uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (batch.t[mid].first_chunk <= chunk) lo = mid;
        else hi = mid - 1;
    }
    const DsAdamwTensor T = batch.t[lo];
    const long long base = (chunk - T.first_chunk) * AW_CHUNK;
    const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2], gs = hyper[3];
    const bool al = ((((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0);
    if (al && base + AW_CHUNK <= T.n) {
        // a whole chunk (all but the last of a tensor): its sixteen 16-byte loads per thread are issued before the first update
        constexpr int NI = AW_CHUNK / 1024;
        f32x4 p[NI], g[NI], m[NI], v[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
            // (streamed once per iteration: non-temporal accesses, 5.7 -> 6.1 TB/s in tools/adamw_bench.py)
            p[it] = __builtin_nontemporal_load((const f32x4*)(T.p + i)); g[it] = __builtin_nontemporal_load((const f32x4*)(T.g + i));
            m[it] = __builtin_nontemporal_load((const f32x4*)(T.m + i)); v[it] = __builtin_nontemporal_load((const f32x4*)(T.v + i));
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[it][e], me = m[it][e], ve = v[it][e];
                aw_update(pe, g[it][e], me, ve, lr, bc1, bc2s, gs, b1, b2, eps, wd);
                p[it][e] = pe; m[it][e] = me; v[it][e] = ve;
            }
            __builtin_nontemporal_store(p[it], (f32x4*)(T.p + i));
            __builtin_nontemporal_store(m[it], (f32x4*)(T.m + i));
            __builtin_nontemporal_store(v[it], (f32x4*)(T.v + i));
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < AW_CHUNK / 1024; ++it) {
        const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
        if (i >= T.n) break;
        if (al && i + 4 <= T.n) {
            f32x4 p = *(const f32x4*)(T.p + i), g = *(const f32x4*)(T.g + i), m = *(const f32x4*)(T.m + i), v = *(const f32x4*)(T.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                aw_update(pe, g[e], me, ve, lr, bc1, bc2s, gs, b1, b2, eps, wd);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            *(f32x4*)(T.p + i) = p;
            *(f32x4*)(T.m + i) = m;
            *(f32x4*)(T.v + i) = v;
        } else {
            for (int e = 0; e < 4 && i + e < T.n; ++e)
                aw_update(T.p[i + e], T.g[i + e], T.m[i + e], T.v[i + e], lr, bc1, bc2s, gs, b1, b2, eps, wd);
        }
    }
}

// tensors: HOST array of n_tensors records { p, g, m, v (device pointers), n (int64 elements) } = 5 x 8 bytes each.
extern "C" int ds_adamw_multi(const void* tensors, int n_tensors, const float* hyper, float beta1, float beta2, float eps,
                              float weight_decay, ds_stream_t stream) {
    DS_CHECK_ARG(tensors && hyper && n_tensors > 0, "bad arguments");
    const long long* rec = (const long long*)tensors;
    for (int i0 = 0; i0 < n_tensors; i0 += AW_BATCH) {
        DsAdamwBatch b;
        b.n = n_tensors - i0 < AW_BATCH ? n_tensors - i0 : AW_BATCH;
        long long chunks = 0;
        for (int j = 0; j < b.n; ++j) {
            const long long* r = rec + (size_t)(i0 + j) * 5;
            DS_CHECK_ARG(r[0] && r[1] && r[2] && r[3] && r[4] > 0, "null pointer / empty tensor in the table");
            b.t[j].p = (float*)(uintptr_t)r[0];
            b.t[j].g = (const float*)(uintptr_t)r[1];
            b.t[j].m = (float*)(uintptr_t)r[2];
            b.t[j].v = (float*)(uintptr_t)r[3];
            b.t[j].n = r[4];
            b.t[j].first_chunk = chunks;
            chunks += (r[4] + AW_CHUNK - 1) / AW_CHUNK;
        }
        DS_CHECK_ARG(chunks < (1ll << 31), "too many chunks in one batch");
        hipLaunchKernelGGL(ds_adamw_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, b, hyper, beta1, beta2,
                           eps, weight_decay);
        DS_CHECK_LAUNCH();
    }
    return 0;
}

// ---- multi-tensor EMA (engine/ema.py:40-56: ema = ema * decay + current * (1 - decay) over the whole state dict) ------------------
// The reference's loop is three elementwise launches per tensor, 767 tensors: ~25 ms every `update_interval` iterations when the
// average lives on the GPU.  Same by-value descriptor batches as ds_adamw_multi: 12 launches, one pass over the bytes.
struct DsEmaTensor {
    float* e;
    const float* c;
    long long n, first_chunk;
};
#define EMA_BATCH 96
struct DsEmaBatch {
    DsEmaTensor t[EMA_BATCH];
    int n;
};
__global__ __launch_bounds__(256) void ds_ema_multi_kernel(const DsEmaBatch batch, float decay, float w) {
#pragma clang fp contract(off)      // two products and one sum, each rounded: never a fused multiply-add (hipcc contracts by default)
    const long long chunk = blockIdx.x;
    int lo = 0, hi = batch.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batch.t[mid].first_chunk <= chunk) lo = mid;
        else hi = mid - 1;
    }
    const DsEmaTensor T = batch.t[lo];
    const long long base = (chunk - T.first_chunk) * AW_CHUNK;
    const bool al = ((((uintptr_t)T.e | (uintptr_t)T.c) & 15) == 0);
#pragma unroll
    for (int it = 0; it < AW_CHUNK / 1024; ++it) {
        const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
        if (i >= T.n) break;
        if (al && i + 4 <= T.n) {
            const f32x4 e = *(const f32x4*)(T.e + i), c = *(const f32x4*)(T.c + i);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                 // two products, one sum, each rounded -- as torch's three elementwise ops
                const float a = e[k] * decay, b = c[k] * w;  // do (the pragma above: no FMA; __fmul_rn / __fadd_rn are plain
                o[k] = a + b;                                // operators in HIP's headers and DO get contracted)
            }
            *(f32x4*)(T.e + i) = o;
        } else {
            for (int k = 0; k < 4 && i + k < T.n; ++k) {
                const float a = T.e[i + k] * decay, b = T.c[i + k] * w;
                T.e[i + k] = a + b;
            }
        }
    }
}

// tensors: HOST array of n_tensors records { ema, current (device pointers, fp32), n (int64 elements) } = 3 x 8 bytes each.
// one_minus_decay: the host's (1 - decay) rounded to fp32 -- what the reference's `cur * (1 - self.decay)` multiplies by (1.f - decay
// evaluated in fp32 is a different number)
extern "C" int ds_ema_multi(const void* tensors, int n_tensors, float decay, float one_minus_decay, ds_stream_t stream) {
    DS_CHECK_ARG(tensors && n_tensors > 0, "bad arguments");
    const long long* rec = (const long long*)tensors;
    for (int i0 = 0; i0 < n_tensors; i0 += EMA_BATCH) {
        DsEmaBatch b;
        b.n = n_tensors - i0 < EMA_BATCH ? n_tensors - i0 : EMA_BATCH;
        long long chunks = 0;
        for (int j = 0; j < b.n; ++j) {
            const long long* r = rec + (size_t)(i0 + j) * 3;
            DS_CHECK_ARG(r[0] && r[1] && r[2] > 0, "null pointer / empty tensor in the table");
            b.t[j].e = (float*)(uintptr_t)r[0];
            b.t[j].c = (const float*)(uintptr_t)r[1];
            b.t[j].n = r[2];
            b.t[j].first_chunk = chunks;
            chunks += (r[2] + AW_CHUNK - 1) / AW_CHUNK;
        }
        DS_CHECK_ARG(chunks < (1ll << 31), "too many chunks in one batch");
        hipLaunchKernelGGL(ds_ema_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, b, decay, one_minus_decay);
        DS_CHECK_LAUNCH();
    }
    return 0;
}

// ---- global gradient norm + clip coefficient (engine/clip_grad_norm.py:8-29 -> torch.nn.utils.clip_grad_norm_) ---------------------
// total = || all gradients ||_2 and coef = min(1, max_norm / (total + 1e-6)), written to device memory for ds_adamw_multi's
// hyper[3]: one pass over the gradient bytes in the same by-value descriptor batches (4096-element chunks, one double per chunk)
// + one small fixed-order reduction -- instead of torch._foreach_norm + stack + vector_norm + clamp + copy (a dozen launches).
struct DsNormTensor {
    const float* g;
    long long n, first_chunk;
};
#define GN_BATCH 128
struct DsNormBatch {
    DsNormTensor t[GN_BATCH];
    int n;
};
__global__ __launch_bounds__(256) void ds_sqnorm_multi_kernel(const DsNormBatch batch, double* __restrict__ part) {
    __shared__ double ws[4];
    const long long chunk = blockIdx.x;
    int lo = 0, hi = batch.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batch.t[mid].first_chunk <= chunk) lo = mid;
        else hi = mid - 1;
    }
    const DsNormTensor T = batch.t[lo];
    const long long base = (chunk - T.first_chunk) * AW_CHUNK;
    const bool al = (((uintptr_t)T.g & 15) == 0);
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < AW_CHUNK / 1024; ++it) {
        const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
        if (i >= T.n) break;
        if (al && i + 4 <= T.n) {
            const f32x4 g = *(const f32x4*)(T.g + i);
            acc += (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
        } else {
            for (int k = 0; k < 4 && i + k < T.n; ++k) acc += T.g[i + k] * T.g[i + k];
        }
    }
    double d = (double)acc;                       // 16 values per thread in fp32, everything above in double
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) part[chunk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(1024) void ds_norm_finish_kernel(const double* __restrict__ part, long long n, float max_norm,
                                                              float* __restrict__ total, float* __restrict__ coef) {
    __shared__ double ws[16];
    double d = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) d += part[i];       // fixed order: bit-reproducible
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += ws[w];
        const float t = (float)sqrt(s);
        *total = t;
        if (coef) {
            const float c = max_norm / (t + 1e-6f);                       // torch: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
            *coef = max_norm > 0.f ? (c < 1.f ? c : 1.f) : 1.f;
        }
    }
}

// tensors: HOST array of n_tensors records { g (device pointer, fp32), n (int64 elements) } = 2 x 8 bytes each.  part: >= sum of
// ceil(n_i / 4096) doubles of workspace.  total: float[1]; coef (may be null): float[1] = min(1, max_norm / (total + 1e-6)), 1 if
// max_norm <= 0.
extern "C" int ds_grad_norm_multi(const void* tensors, int n_tensors, double* part, long long part_len, float max_norm, float* total,
                                  float* coef, ds_stream_t stream) {
    DS_CHECK_ARG(tensors && n_tensors > 0 && part && total, "bad arguments");
    const long long* rec = (const long long*)tensors;
    long long done = 0;
    for (int i0 = 0; i0 < n_tensors; i0 += GN_BATCH) {
        DsNormBatch b;
        b.n = n_tensors - i0 < GN_BATCH ? n_tensors - i0 : GN_BATCH;
        long long chunks = 0;
        for (int j = 0; j < b.n; ++j) {
            const long long* r = rec + (size_t)(i0 + j) * 2;
            DS_CHECK_ARG(r[0] && r[1] > 0, "null pointer / empty tensor in the table");
            b.t[j].g = (const float*)(uintptr_t)r[0];
            b.t[j].n = r[1];
            b.t[j].first_chunk = chunks;
            chunks += (r[1] + AW_CHUNK - 1) / AW_CHUNK;
        }
        DS_CHECK_ARG(chunks < (1ll << 31) && done + chunks <= part_len, "workspace too small for the gradients' chunks");
        hipLaunchKernelGGL(ds_sqnorm_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, b, part + done);
        DS_CHECK_LAUNCH();
        done += chunks;
    }
    hipLaunchKernelGGL(ds_norm_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, done, max_norm, total, coef);
    DS_CHECK_LAUNCH();
    return 0;
}
